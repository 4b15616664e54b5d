// twhost.cpp — see twhost.h.  Host-side only: file decode, size reconcile, queueing.  All pixel arithmetic of
// the hot path happens in libtwflow.so (HIP); there is no CPU flow implementation here.
#include "twhost.h"
#include "tw_inflate.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>

namespace twhost {

// ---------------------------------------------------------------------------------------------------
// progress prints of the reference (std::cout in src/manager.cpp:74, src/consumer.cpp:49,55,92), behind TW_LOG
// (the reference's own CLI had to dup2 stdout away to hide them).  TW_LOG=1: stdout like the
// reference, TW_LOG=2: stderr.
// ---------------------------------------------------------------------------------------------------
static FILE* log_stream()
{
    static FILE* f = [] {
        const char* ev = getenv("TW_LOG");
        const int v = ev ? atoi(ev) : 0;
        return v == 1 ? stdout : (v >= 2 ? stderr : (FILE*)nullptr);
    }();
    return f;
}
#define TW_LOGF(...)                      \
    do {                                  \
        if (FILE* lf_ = log_stream()) {   \
            fprintf(lf_, __VA_ARGS__);    \
            fflush(lf_);                  \
        }                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// NUMA placement (SURVEY.md 8(e) "scaling risks": pin each worker + its staging buffers to the GPU's NUMA node)
// ---------------------------------------------------------------------------------------------------
static std::string sysfs_root()
{
    const char* ev = getenv("TW_SYSFS_ROOT");
    return ev && ev[0] ? std::string(ev) : std::string("/sys");
}

static bool read_first_line(const std::string& path, std::string& out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, out);
    return true;
}

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}
static std::vector<int> parse_cpulist(const std::string& s)
{
    std::vector<int> v;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        if (tok.empty()) continue;
        const size_t dash = tok.find('-');
        const int a = atoi(tok.c_str());
        const int b = dash == std::string::npos ? a : atoi(tok.c_str() + dash + 1);
        for (int c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) v.push_back(c);
    }
    return v;
}

bool numa_cpus_of_device(int device, std::string* busid, int* node, std::vector<int>* cpus)
{
    char id[32];
    if (tw_device_pci_bus_id(device, id, (int)sizeof(id)) != TW_OK || !id[0]) return false;
    if (busid) *busid = id;
    std::string line;
    if (!read_first_line(sysfs_root() + "/bus/pci/devices/" + id + "/numa_node", line)) return false;
    const int n = atoi(line.c_str());
    if (node) *node = n;
    if (n < 0) return false;  // the platform reports no NUMA affinity for this device
    if (!read_first_line(sysfs_root() + "/devices/system/node/node" + std::to_string(n) + "/cpulist", line)) return false;
    std::vector<int> v = parse_cpulist(line);
    if (v.empty()) return false;
    if (cpus) *cpus = std::move(v);
    return true;
}

std::vector<int> this_thread_cpus()
{
    std::vector<int> v;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) != 0) return v;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &set)) v.push_back(c);
    return v;
}

bool bind_this_thread(const std::vector<int>& cpus)
{
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    int n = 0;
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
            CPU_SET(c, &want);
            n++;
        }
    if (n == 0) return false;  // e.g. a container whose CPU share lies on another node: stay where we are
    return sched_setaffinity(0, sizeof(want), &want) == 0;
}

// ---------------------------------------------------------------------------------------------------
// image files -> 8-bit gray  (cv::imread(path, IMREAD_GRAYSCALE), src/opticalflow.cpp:37,44)
// ---------------------------------------------------------------------------------------------------
static bool read_file(const std::string& path, std::vector<uint8_t>& buf)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) {
        fclose(f);
        return false;
    }
    buf.resize((size_t)n);
    const size_t got = fread(buf.data(), 1, (size_t)n, f);
    fclose(f);
    return got == (size_t)n;
}

// OpenCV 2.4.9's own decoders (BMP, PxM) convert colour to gray with icvCvt_BGR2Gray_8u_C3C1R (highgui/utils.cpp):
// 14-bit coefficients, rounded — NOT libpng's formula.  No fixture of the reference is a BMP or a PNM: parity unpinned.
static inline uint8_t bgr_to_gray_cv(int b, int g, int r)
{
    return (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
}

// PBM / PGM / PPM, binary and ASCII (P1-P6), maxval <= 255 (highgui/grfmt_pxm.cpp: values are scaled by 255 / maxval,
// bitmaps read 1 as black)
static bool decode_pnm(const std::vector<uint8_t>& d, std::vector<uint8_t>& img, int& w, int& h)
{
    const int kind = d[1] - '0';
    if (kind < 1 || kind > 6) return false;
    size_t p = 2;
    auto skip = [&] {
        for (;;) {
            while (p < d.size() && (d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t')) p++;
            if (p < d.size() && d[p] == '#') {
                while (p < d.size() && d[p] != '\n') p++;
                continue;
            }
            return;
        }
    };
    auto number = [&](int& v) -> bool {
        skip();
        long long acc = 0;
        int nd = 0;
        while (p < d.size() && d[p] >= '0' && d[p] <= '9' && nd < 10) {
            acc = acc * 10 + (d[p] - '0');
            p++;
            nd++;
        }
        if (!nd || acc > 0x7fffffff) return false;
        v = (int)acc;
        return true;
    };
    const bool bitmap = kind == 1 || kind == 4, colour = kind == 3 || kind == 6, ascii = kind <= 3;
    int maxval = 1;
    if (!number(w) || !number(h) || (!bitmap && !number(maxval))) return false;
    if (w < 1 || h < 1 || w > 32768 || h > 32768 || maxval < 1 || maxval > 255) return false;
    if (!ascii) p++;  // a single whitespace byte ends the header of the binary kinds
    const size_t npx = (size_t)w * h;
    {
        // the payload must be able to hold the image the header announces BEFORE anything of that size is allocated
        // (a 20-byte file with a 32768 x 32768 header used to cost 1 GiB of zero-filled memory per decode
        // thread): binary kinds exactly; ASCII kinds at least one character per sample (P1) or a digit and a
        // separator per sample but the last (P2 / P3)
        const size_t rest = p < d.size() ? d.size() - p : 0;
        const size_t samples = npx * (colour ? 3 : 1);
        const size_t least = !ascii ? (bitmap ? (((size_t)w + 7) / 8) * (size_t)h : samples) : bitmap ? samples : 2 * samples - 1;
        if (rest < least) return false;
    }
    img.resize(npx);
    uint8_t scale[256];
    for (int i = 0; i <= maxval; i++) scale[i] = (uint8_t)(i * 255 / maxval);
    if (ascii) {
        for (size_t i = 0; i < npx; i++) {
            int v[3] = {0, 0, 0};
            for (int c = 0; c < (colour ? 3 : 1); c++) {
                if (bitmap) {  // digits need no separators in P1
                    skip();
                    if (p >= d.size() || (d[p] != '0' && d[p] != '1')) return false;
                    v[c] = d[p++] - '0';
                } else if (!number(v[c]) || v[c] > maxval) {
                    return false;
                }
            }
            img[i] = bitmap ? (v[0] ? 0 : 255) : colour ? bgr_to_gray_cv(scale[v[2]], scale[v[1]], scale[v[0]]) : scale[v[0]];
        }
        return true;
    }
    if (bitmap) {
        const size_t rb = ((size_t)w + 7) / 8;
        if (p + rb * h > d.size()) return false;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (d[p + rb * y + (x >> 3)] >> (7 - (x & 7))) & 1 ? 0 : 255;
        return true;
    }
    const size_t need = npx * (colour ? 3 : 1);
    if (p + need > d.size()) return false;
    const uint8_t* q = &d[p];
    for (size_t i = 0; i < npx; i++) {
        if (colour) {
            if (q[3 * i] > maxval || q[3 * i + 1] > maxval || q[3 * i + 2] > maxval) return false;
            img[i] = bgr_to_gray_cv(scale[q[3 * i + 2]], scale[q[3 * i + 1]], scale[q[3 * i]]);
        } else {
            if (q[i] > maxval) return false;
            img[i] = scale[q[i]];
        }
    }
    return true;
}

// Windows / OS/2 bitmaps, uncompressed: 1 / 4 / 8 bits with a palette, 24 and 32 bits (highgui/grfmt_bmp.cpp reads RLE4 /
// RLE8 and 16-bit files as well: those answer "Can't open" here)
static bool decode_bmp(const std::vector<uint8_t>& d, std::vector<uint8_t>& img, int& w, int& h)
{
    auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (d.size() < 26) return false;
    const uint32_t off = u32(10), hs = u32(14);
    int bpp, comp = 0;
    bool topdown = false;
    size_t pal_entry = 4;
    if (hs == 12) {  // OS/2 core header: 16-bit sizes, 3-byte palette entries
        w = (int)u16(18);
        h = (int)u16(20);
        bpp = (int)u16(24);
        pal_entry = 3;
    } else if (hs >= 40 && d.size() >= 14 + 40) {
        w = (int)u32(18);
        int hh = (int)u32(22);
        topdown = hh < 0;
        h = topdown ? -hh : hh;
        bpp = (int)u16(28);
        comp = (int)u32(30);
    } else {
        return false;
    }
    if (w < 1 || h < 1 || w > 32768 || h > 32768) return false;
    if (comp != 0 || !(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32)) return false;
    uint8_t pal[256];
    if (bpp <= 8) {
        const size_t po = 14 + (size_t)hs;
        uint32_t ncol = hs >= 40 ? u32(46) : 0;
        if (ncol == 0 || ncol > (1u << bpp)) ncol = 1u << bpp;
        if (po + pal_entry * ncol > d.size()) return false;
        memset(pal, 0, sizeof(pal));
        for (uint32_t i = 0; i < ncol; i++) pal[i] = bgr_to_gray_cv(d[po + pal_entry * i], d[po + pal_entry * i + 1], d[po + pal_entry * i + 2]);
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)off + stride * h > d.size() || off < 14 + hs) return false;
    img.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = &d[(size_t)off + stride * (size_t)(topdown ? y : h - 1 - y)];
        uint8_t* out = &img[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            if (bpp == 24) out[x] = bgr_to_gray_cv(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
            else if (bpp == 32) out[x] = bgr_to_gray_cv(row[4 * x], row[4 * x + 1], row[4 * x + 2]);
            else if (bpp == 8) out[x] = pal[row[x]];
            else if (bpp == 4) out[x] = pal[(row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15];
            else out[x] = pal[(row[x >> 3] >> (7 - (x & 7))) & 1];
        }
    }
    return true;
}

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// libpng 1.5.12 (the copy OpenCV 2.4.9 bundles) png_set_rgb_to_gray(…, 0.299, 0.587): truncated 15-bit
// coefficients and a truncated sum; pinned by the reference's golden vectors (tests/golden/make_fixtures.py).
static inline uint8_t rgb_to_gray(int r, int g, int b)
{
    if (r == g && g == b) return (uint8_t)r;
    return (uint8_t)((9797 * r + 19234 * g + 3737 * b) >> 15);
}

// CRC-32 of PNG chunks (ISO 3309, reflected 0xEDB88320), slicing-by-8: ~1 % of a decode
static uint32_t png_crc32(const uint8_t* p, size_t n)
{
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int t = 1; t < 8; t++) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
    });
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = T[7][a & 0xFF] ^ T[6][(a >> 8) & 0xFF] ^ T[5][(a >> 16) & 0xFF] ^ T[4][a >> 24] ^ T[3][b & 0xFF] ^
            T[2][(b >> 8) & 0xFF] ^ T[1][(b >> 16) & 0xFF] ^ T[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// What is left of a PNG after its chunks are parsed and its IDAT stream is inflated: the filtered scanlines (one pass,
// or the seven Adam7 passes one after the other) and what is needed to finish them
struct PngStream {
    std::unique_ptr<uint8_t[]> raw;  // `total` bytes of filtered rows (+ 8 bytes of slack)
    size_t total = 0;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0, ch = 0;
    std::vector<uint8_t> plte, idat;  // idat: the concatenated IDAT chunks between png_parse and png_inflate
    // 8-bit, not interlaced, gray / gray + alpha / RGB / RGBA: rows the device reconstructs (tw_submit_png8)
    bool device_rows() const { return depth == 8 && !interlace && ctype != 3; }
};

// stage 1a of cv::imread on a PNG: the chunk walk (libpng's rules); the IDAT stream is gathered in ps.idat
static bool png_parse(const std::vector<uint8_t>& d, PngStream& ps)
{
    int w = 0, h = 0;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t>& idat = ps.idat;
    std::vector<uint8_t>& plte = ps.plte;
    bool have_ihdr = false, idat_done = false, first = true, have_iend = false;
    // Chunk rules the way libpng (cv::imread) enforces them — a file it refuses answers "Can't open" here as well
    // (ADVICE r3): IHDR first; the CRC of every CRITICAL chunk (IHDR, PLTE, IDAT; an upper-case first letter) must
    // hold — a mismatch there is png_error, in an ancillary chunk only a warning; IDAT chunks are consecutive; PLTE
    // comes before IDAT.  ADVICE r4 (parity unpinned — no fixture of the reference is such a file): IHDR is exactly 13
    // bytes with compression method 0 and filter method 0 (png_check_IHDR); a critical chunk libpng does not know is
    // png_chunk_error; IEND is a critical chunk too (its CRC must hold) and must be there — cv::imread calls
    // png_read_end inside its setjmp block, so a file that stops after its last IDAT fails to open.
    while (p + 12 <= d.size()) {
        const uint32_t len = be32(&d[p]);
        const char* type = (const char*)&d[p + 4];
        if (p + 12 + (size_t)len > d.size()) return false;
        const uint8_t* data = &d[p + 8];
        const bool critical = !(type[0] & 0x20);
        if (critical && png_crc32(&d[p + 4], 4 + (size_t)len) != be32(data + len)) return false;
        if (first && memcmp(type, "IHDR", 4) != 0) return false;
        first = false;
        const bool is_idat = !memcmp(type, "IDAT", 4);
        if (!is_idat && !idat.empty()) idat_done = true;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) return false;
            if (data[10] != 0 || data[11] != 0) return false;  // compression method, filter method
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            if (!idat.empty()) return false;
            plte.assign(data, data + len);
        } else if (is_idat) {
            if (idat_done) return false;  // IDATs must be consecutive
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            have_iend = true;
            break;
        } else if (critical) {
            return false;  // unknown critical chunk
        }
        p += 12 + (size_t)len;
    }
    if (!have_iend) return false;
    if (!have_ihdr || w < 1 || h < 1 || w > 32768 || h > 32768 || interlace > 1) return false;
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return false;
    const size_t bpp_bits = (size_t)ch * depth;
    // Adam7 (interlace 1): seven reduced images one after the other, each with its own filtered rows; a pass is
    // unfiltered and converted like a whole image and its pixels scattered to (x0 + i*dx, y0 + j*dy)
    static const int AX0[7] = {0, 4, 0, 2, 0, 1, 0}, AY0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int ADX[7] = {8, 8, 4, 4, 2, 2, 1}, ADY[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    size_t total = 0;
    for (int ps = 0; ps < npass; ps++) {
        const int pw = interlace ? (w - AX0[ps] + ADX[ps] - 1) / ADX[ps] : w;
        const int ph = interlace ? (h - AY0[ps] + ADY[ps] - 1) / ADY[ps] : h;
        if (pw > 0 && ph > 0) total += (((size_t)pw * bpp_bits + 7) / 8 + 1) * (size_t)ph;
    }
    // DEFLATE expands by at most 1032 : 1: a stream too short for the image the header announces is refused before
    // `total` bytes (up to 8 GiB for 32768 x 32768 RGBA16) are allocated for it
    if (total / 1032 > idat.size()) return false;
    ps.total = total;
    ps.w = w;
    ps.h = h;
    ps.depth = depth;
    ps.ctype = ctype;
    ps.interlace = interlace;
    ps.ch = ch;
    return true;
}

// stage 1b: the inflate, into `dst` (ps.total bytes + 8 of slack: the caller's arena slot, or ps.raw)
static bool png_inflate(const PngStream& ps, uint8_t* dst)
{
    size_t outlen = 0;
    return tw_inflate_zlib(ps.idat.data(), ps.idat.size(), dst, ps.total, &outlen) && outlen == ps.total;
}

static bool png_parse_inflate(const std::vector<uint8_t>& d, PngStream& ps)
{
    if (!png_parse(d, ps)) return false;
    // (uninitialised: every byte is written by the inflate or the decode fails; + 8 bytes of slack for nothing — the
    // decoder never writes past `total`)
    ps.raw.reset(new (std::nothrow) uint8_t[ps.total + 8]);
    if (!ps.raw) return false;
    const bool ok = png_inflate(ps, ps.raw.get());
    std::vector<uint8_t>().swap(ps.idat);
    return ok;
}

// stage 2: scanline reconstruction (tw_inflate.cpp) + conversion to 8-bit gray with libpng 1.5's formula — on the host
// (every PNG kind; what tw_png_unfilter does on the device for the 8-bit non-interlaced kinds)
static bool png_finish_on_host(PngStream& st, std::vector<uint8_t>& img)
{
    uint8_t* const raw = st.raw.get();
    const int w = st.w, h = st.h, depth = st.depth, ctype = st.ctype, interlace = st.interlace, ch = st.ch;
    const std::vector<uint8_t>& plte = st.plte;
    const size_t bpp_bits = (size_t)ch * depth;
    const size_t fbpp = std::max<size_t>(1, bpp_bits / 8);
    static const int AX0[7] = {0, 4, 0, 2, 0, 1, 0}, AY0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int ADX[7] = {8, 8, 4, 4, 2, 2, 1}, ADY[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    img.resize((size_t)w * h);
    size_t pass_off = 0;
    const int W = w, H = h;  // the loops below run over one pass: w/h are its dimensions
    for (int ps = 0; ps < npass; ps++) {
    const int x0 = interlace ? AX0[ps] : 0, y0 = interlace ? AY0[ps] : 0;
    const int dx = interlace ? ADX[ps] : 1, dy = interlace ? ADY[ps] : 1;
    const int w = interlace ? (W - x0 + dx - 1) / dx : W, h = interlace ? (H - y0 + dy - 1) / dy : H;
    if (w <= 0 || h <= 0) continue;
    const size_t rowbytes = ((size_t)w * bpp_bits + 7) / 8;
    std::vector<uint8_t> line((size_t)w);
    // unfilter the whole pass in place (tw_inflate.cpp: one loop per filter type, Paeth rows of 1-byte pixels two at a
    // time), then convert row by row
    if (!tw_png_unfilter(raw + pass_off, rowbytes, (size_t)h, fbpp)) return false;
    for (int y = 0; y < h; y++) {
        uint8_t* row = &raw[pass_off + (rowbytes + 1) * (size_t)y];
        uint8_t* cur = row + 1;
        uint8_t* out = line.data();
        if (depth == 8 && ctype == 0 && dx == 1) {
            memcpy(&img[(size_t)(y0 + y * dy) * W + x0], cur, (size_t)w);  // 8-bit gray, not interlaced: the row is the output
            continue;
        }
        auto sample8 = [&](size_t idx) -> int {  // idx-th sample of the row as 8 bits (16-bit: high byte)
            if (depth == 8) return cur[idx];
            if (depth == 16) return cur[idx * 2];
            const int per = 8 / depth;
            const int v2 = (cur[idx / per] >> ((per - 1 - (int)(idx % per)) * depth)) & ((1 << depth) - 1);
            return ctype == 3 ? v2 : v2 * 255 / ((1 << depth) - 1);
        };
        if (depth == 8 && ctype == 0) {
            memcpy(out, cur, (size_t)w);  // 8-bit gray: the row is the output
        } else if (depth == 8 && (ctype == 2 || ctype == 6)) {
            for (int x = 0; x < w; x++) out[x] = rgb_to_gray(cur[(size_t)x * ch], cur[(size_t)x * ch + 1], cur[(size_t)x * ch + 2]);
        } else
        for (int x = 0; x < w; x++) {
            if (ctype == 0 || ctype == 4) {
                out[x] = (uint8_t)sample8((size_t)x * ch);
            } else if (ctype == 3) {
                const size_t idx = (size_t)sample8((size_t)x) * 3;
                if (idx + 2 >= plte.size()) return false;
                out[x] = rgb_to_gray(plte[idx], plte[idx + 1], plte[idx + 2]);
            } else {
                out[x] = rgb_to_gray(sample8((size_t)x * ch), sample8((size_t)x * ch + 1), sample8((size_t)x * ch + 2));
            }
        }
        uint8_t* dst = &img[(size_t)(y0 + y * dy) * W + x0];
        for (int x = 0; x < w; x++) dst[(size_t)x * dx] = out[x];
    }
    pass_off += (rowbytes + 1) * (size_t)h;
    }
    return true;
}

static bool decode_png(const std::vector<uint8_t>& d, std::vector<uint8_t>& img, int& w, int& h)
{
    PngStream st;
    if (!png_parse_inflate(d, st)) return false;
    w = st.w;
    h = st.h;
    return png_finish_on_host(st, img);
}

// The rest of the decode of rows load_gray_or_png_rows() returned (size reconcile needs gray pixels on the host)
bool finish_png_rows_on_host(std::vector<uint8_t>& rows, int w, int h, int ch, std::vector<uint8_t>& gray)
{
    if (ch < 1 || ch > 4 || rows.size() != (size_t)h * ((size_t)w * ch + 1)) return false;
    PngStream st;
    st.total = rows.size();
    st.raw.reset(new (std::nothrow) uint8_t[st.total + 8]);
    if (!st.raw) return false;
    memcpy(st.raw.get(), rows.data(), st.total);
    st.w = w;
    st.h = h;
    st.depth = 8;
    st.interlace = 0;
    st.ch = ch;
    st.ctype = ch == 1 ? 0 : ch == 2 ? 4 : ch == 3 ? 2 : 6;
    return png_finish_on_host(st, gray);
}

static bool decode_gray_or_png_rows(const std::vector<uint8_t>& d, bool want_rows, std::vector<uint8_t>& img, int& w, int& h, int& ch);

bool load_gray(const std::string& path, std::vector<uint8_t>& img, int& w, int& h)
{
    int ch = 0;
    return load_gray_or_png_rows(path, false, img, w, h, ch);
}

// cv::imread(path, GRAYSCALE) — or, with want_rows and an 8-bit non-interlaced gray / gray + alpha / RGB / RGBA PNG,
// only its first half: `img` then holds the inflated, still FILTERED scanlines (h rows of 1 + w * ch bytes) and ch is
// 1-4; the device finishes them (tw_submit_png8).  Every row's filter type is checked here (0-4, as libpng does).
// ch = 0: `img` is the gray image.
bool load_gray_or_png_rows(const std::string& path, bool want_rows, std::vector<uint8_t>& img, int& w, int& h, int& ch)
{
    std::vector<uint8_t> d;
    w = h = ch = 0;
    if (!read_file(path, d)) return false;
    return decode_gray_or_png_rows(d, want_rows, img, w, h, ch);
}

// the same on the bytes of a file that has been read already
static bool decode_gray_or_png_rows(const std::vector<uint8_t>& d, bool want_rows, std::vector<uint8_t>& img, int& w, int& h, int& ch)
{
    w = h = ch = 0;
    if (d.size() < 8) return false;
    bool ok = false;
    if (d[0] == 'P' && d[1] >= '1' && d[1] <= '6') ok = decode_pnm(d, img, w, h);
    else if (d[0] == 0x89 && d[1] == 'P' && want_rows) {
        PngStream st;
        ok = png_parse_inflate(d, st);
        if (ok) {
            w = st.w;
            h = st.h;
            if (st.device_rows()) {
                const size_t rs = (size_t)st.w * st.ch + 1;
                for (int y = 0; y < st.h && ok; y++) ok = st.raw[(size_t)y * rs] <= 4;
                if (ok) {
                    img.assign(st.raw.get(), st.raw.get() + st.total);
                    ch = st.ch;
                }
            } else {
                ok = png_finish_on_host(st, img);
            }
        }
    } else if (d[0] == 0x89 && d[1] == 'P') ok = decode_png(d, img, w, h);
    else if (d[0] == 0xFF && d[1] == 0xD8) ok = decode_jpeg_gray(d.data(), d.size(), img, w, h);
    else if (d[0] == 'B' && d[1] == 'M') ok = decode_bmp(d, img, w, h);
    // (the other cv::imread formats — TIFF, JPEG-2000, Sun raster, ... — are not decoded: INTEGRATION.md)
    if (!ok) {
        // the decoders write w / h from the header before they can fail: a failed decode reports no size (ADVICE r3:
        // a 32768 x 32768 IHDR over a truncated IDAT used to size the page-locked arena)
        w = h = ch = 0;
        std::vector<uint8_t>().swap(img);
    }
    return ok;
}

// cv::resize on CV_8UC1, INTER_LINEAR, 11-bit fixed point (imgproc/imgwarp.cpp: HResizeLinear<uchar,int,short>,
// VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>).  No golden vector of the reference exercises
// this branch (its fixtures have equal sizes): parity unpinned.
void resize_u8_linear(const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst, int dw, int dh)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ia(2 * (size_t)dw), ib(2 * (size_t)dh);
    auto sat_short = [](double v) { long r = lrint(v); return (short)std::min<long>(32767, std::max<long>(-32768, r)); };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ia[2 * dx] = sat_short((1.f - fx) * 2048);
        ia[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ib[2 * dy] = sat_short((1.f - fy) * 2048);
        ib[2 * dy + 1] = sat_short(fy * 2048);
    }
    dst.resize((size_t)dw * dh);
    std::vector<int> r0(dw), r1(dw);
    auto hrow = [&](int sy, std::vector<int>& out) {
        sy = std::min(std::max(sy, 0), sh - 1);
        const uint8_t* S = &src[(size_t)sy * sw];
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx];
            out[dx] = sx + 1 < sw ? S[sx] * ia[2 * dx] + S[sx + 1] * ia[2 * dx + 1] : S[sx] * 2048;
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        hrow(yofs[dy], r0);
        hrow(yofs[dy] + 1, r1);
        const int b0 = ib[2 * dy], b1 = ib[2 * dy + 1];
        for (int dx = 0; dx < dw; dx++) {
            const int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dw + dx] = (uint8_t)std::min(255, std::max(0, v));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Consumer
// ---------------------------------------------------------------------------------------------------
Consumer::Consumer(int id, MessageQueue<Request>& req, MessageQueue<Response>& res, const tw_params& p, int batch,
                   int decode_threads, int n_consumers, ConsumerShared* shared)
    : id_(id), req_(req), res_(res), params_(p), batch_(std::max(1, batch)), decode_threads_(std::max(1, decode_threads)),
      n_consumers_(std::max(1, n_consumers)), shared_(shared)
{
}
Consumer::~Consumer() { join(); }
void Consumer::start() { th_ = std::thread([this] { run(); }); }
void Consumer::join()
{
    if (th_.joinable()) th_.join();
}

namespace {
inline long long steady_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Staged {
    Request req;
    std::vector<uint8_t> a, b;
    const uint8_t *pa = nullptr, *pb = nullptr;  // the pair as handed to the engine (a/b or the caller's raw buffers)
    ptrdiff_t stride = 0;
    int w = 0, h = 0;
    int cha = 0, chb = 0;  // > 0: a / b hold filtered PNG rows of that many channels (the device finishes them)
    std::string err;
    tw_ticket ticket = 0;
    bool submitted = false;
    bool done = false;  // response pushed
};

// the two imreads of a pair are independent: the decode pool runs them as separate tasks (task = 2 * job + image)
// (`file`: the bytes of the file when the caller has read it already, else null)
void decode_one(Staged& s, int which, int* tw, int* th, std::string* err, bool device_png, const std::vector<uint8_t>* file)
{
    if (s.req.raw.expect) return;
    const std::string& path = which == 0 ? s.req.expect_image : s.req.target_image;
    if (path.empty()) { *err = which == 0 ? "ExpectImagePath is empty." : "TargetImagePath is empty."; return; }
    std::vector<uint8_t>& img = which == 0 ? s.a : s.b;
    int& w = which == 0 ? s.w : *tw;
    int& h = which == 0 ? s.h : *th;
    int& ch = which == 0 ? s.cha : s.chb;
    const bool ok = file ? decode_gray_or_png_rows(*file, device_png, img, w, h, ch)
                         : load_gray_or_png_rows(path, device_png, img, w, h, ch);
    if (!ok) *err = "Can't open " + path;
}

// OpticalFlow::calculate up to (not including) calculateInternal: src/opticalflow.cpp:20-68
void prepare(Staged& s, int tw, int th, const std::string& err_a, const std::string& err_b)
{
    if (s.req.raw.expect) {  // in-memory pair: nothing to decode
        const RawPair& r = s.req.raw;
        if (!r.target || r.width < 1 || r.height < 1 || r.stride < r.width) { s.err = "bad raw image pair"; return; }
        s.pa = r.expect;
        s.pb = r.target;
        s.w = r.width;
        s.h = r.height;
        s.stride = r.stride;
        return;
    }
    // the reference checks and opens the expected image first (src/opticalflow.cpp:26-48): its error wins
    if (s.req.expect_image.empty()) { s.err = "ExpectImagePath is empty."; return; }
    if (s.req.target_image.empty()) { s.err = "TargetImagePath is empty."; return; }
    if (!err_a.empty()) { s.err = err_a; return; }
    if (!err_b.empty()) { s.err = err_b; return; }
    if (abs(s.h - th) > 5 || abs(s.w - tw) > 5) { s.err = "Don't match image size"; return; }
    if (s.h != th || s.w != tw) {
        // the <= 5 px reconcile resizes the TARGET's gray pixels (src/opticalflow.cpp:64-68): a target that is still
        // filtered PNG rows is finished on the host first (rare; the expected image may stay as it is)
        if (s.chb) {
            std::vector<uint8_t> g;
            if (!finish_png_rows_on_host(s.b, tw, th, s.chb, g)) { s.err = "Can't open " + s.req.target_image; return; }
            s.b.swap(g);
            s.chb = 0;
        }
        std::vector<uint8_t> r;
        resize_u8_linear(s.b, tw, th, r, s.w, s.h);
        s.b.swap(r);
    }
    s.pa = s.a.data();
    s.pb = s.b.data();
    s.stride = s.w;
}
}  // namespace

void Consumer::run()
{
    // device binding happens here, on the worker thread (the reference binds on the main thread:
    // src/consumer.cpp:22 via src/manager.cpp:56 — SURVEY.md Appendix B#6)
    const int ndev = tw_device_count();
    const int dev = ndev > 0 ? id_ % ndev : -1;
    ConsumerStats mine;  // this thread's own counters; published under the shared mutex
    mine.id = id_;
    mine.device = dev;
    // NUMA placement BEFORE the engine exists: the thread — and with it the decode pool it spawns and, by first touch,
    // the page-locked staging / result buffers tw_engine_create and tw_submit_u8 allocate — moves to the CPUs of the
    // GPU's node, so that 8 consumers x ~16 GB/s of pinned H2D do not cross the socket link.  TW_NUMA=0 disables.
    {
        const char* ev = getenv("TW_NUMA");
        const bool on = !ev || atoi(ev) != 0;
        std::vector<int> cpus;
        int node = -1;
        if (dev >= 0 && numa_cpus_of_device(dev, &mine.pciBusId, &node, &cpus) && on && bind_this_thread(cpus))
            mine.numaNode = node;
        mine.cpus = this_thread_cpus();
    }
    // PNG scanline reconstruction + gray conversion on the device (tw_submit_png8): the decode pool only inflates.
    // TW_DEVICE_PNG=0 keeps the whole decode on the host (A/B switch; same bytes either way).
    bool device_png = true;
    if (const char* ev = getenv("TW_DEVICE_PNG")) device_png = atoi(ev) != 0;
    tw_engine* eng = nullptr;
    std::string eng_err;
    if (ndev > 0) {
        tw_status r = tw_engine_create(dev, &params_, batch_, &eng);
        if (r != TW_OK) eng_err = std::string("engine: ") + tw_strerror(r);
    } else {
        eng_err = "no HIP device available";
    }
    // Opt-in (TW_SCAN_FUSED_FINAL=1): the engine option that evaluates the last level-0 iteration at the span-grid points
    // only.  A consumer hands back vectors, never the flow field, so the option is invisible to the caller (bit-identical
    // vectors, tests/test_gpu_parity.py::test_scan_fused_final_iteration_option) and worth +3-5 % of GPU throughput; it
    // stays off by default because the reference's consumer computes the whole field (src/consumer.cpp:54).
    if (eng)
        if (const char* ev = getenv("TW_SCAN_FUSED_FINAL"))
            if (atoi(ev) != 0) (void)tw_set_option(eng, TW_OPT_SCAN_FUSED_FINAL, 1);
    const bool prof = eng && shared_ && shared_->profile;
    if (prof) {
        (void)tw_prof_select(eng, TW_K_BLUR_SOLVE, 0);
        (void)tw_prof_select(eng, TW_K_POLYEXP, 0);
    }
    int my_epoch = shared_ ? shared_->epoch.load() : 0;
    auto publish = [&] {
        if (!shared_) return;
        std::lock_guard<std::mutex> lk(shared_->m);
        if ((size_t)id_ < shared_->stats.size()) shared_->stats[(size_t)id_] = mine;
        shared_->cv.notify_all();
    };
    // kernel events are read (a device synchronise) only when nothing of ours is in flight
    auto read_prof = [&](bool keep) {
        if (!prof) return;
        static const int kc[2] = {TW_K_BLUR_SOLVE, TW_K_POLYEXP};
        for (int i = 0; i < 2; i++) {
            double ms = 0;
            int n = 0;
            if (tw_prof_read(eng, kc[i], &ms, &n) == TW_OK && keep) {
                mine.profMs[i] += ms;
                mine.profLaunches[i] += n;
            }
        }
    };
    mine.ready = true;
    mine.engineError = eng_err;
    publish();
    // wait for one job and hand its response to the pump
    // (one scratch array for the hit records of a job, grown to the largest grid seen: a fresh zero-filled vector of
    // grid-capacity entries per job — 0.5 MB at 1080p / span 10 — cost 0.2 ms of every response, round 4)
    std::unique_ptr<tw_vector[]> hits;
    size_t hits_cap = 0;
    auto finish = [&](Staged& s) {
        if (s.done) return;
        s.done = true;
        Response res;
        if (s.submitted) {
            const int cap = std::max(1, tw_grid_capacity(s.w, s.h, s.req.span));
            if ((size_t)cap > hits_cap) {
                hits.reset(new tw_vector[(size_t)cap]);
                hits_cap = (size_t)cap;
            }
            tw_vector* const v = hits.get();
            int n = 0;
            float sec = 0;
            tw_status r = tw_wait(eng, s.ticket, v, cap, &n, &sec);
            if (shared_) shared_->inflight--;
            if (r != TW_OK) {
                s.err = std::string(tw_last_error(eng)[0] ? tw_last_error(eng) : tw_strerror(r));
            } else {
                res.vectors.resize((size_t)std::min(n, cap));
                for (size_t i = 0; i < res.vectors.size(); i++) res.vectors[i] = {v[i].x, v[i].y, v[i].dx, v[i].dy};
                res.status = res.vectors.empty() ? "OK" : "SUSPICIOUS";  // src/consumer.cpp:77
                res.time = sec;
                res.expect_image = s.req.expect_image;
                res.target_image = s.req.target_image;
                res.span = s.req.span;
                res.threshold = s.req.threshold;
                res.height = s.h;
                res.width = s.w;
            }
        }
        if (!s.err.empty()) {
            res = Response();
            res.status = "ERROR";  // src/consumer.cpp:85-88
            res.reason = s.err;
        }
        TW_LOGF("finish optical flow: %g\n", (double)res.time);  // src/consumer.cpp:55
        mine.pairs++;
        mine.lastResponseNs = steady_ns();
        publish();  // before the response leaves: whoever has seen N responses sees N pairs in the stats
        res.pushedNs = mine.lastResponseNs;
        res_.push(std::move(res));
    };
    // Two batches are kept going: the one just submitted computes while the next one is popped, decoded and
    // uploaded (the engine itself holds up to three batches).  Responses leave in completion order, as in the
    // reference.
    struct Arena {
        uint8_t* base = nullptr;
        size_t cap = 0;
    } arena[3];  // page-locked (tw_host_alloc): the decoded pairs of the batch being filled / the two batches in flight
    int arena_idx = 0;
    // Batches in flight behind the one being prepared: `prev` was submitted last, `prev2` before it.  Round 4: with one
    // batch in flight the loop ran at the GPU's LATENCY for a batch (upload -> scanline reconstruction -> flow: ~22 ms for
    // 32 pairs) instead of at the host's own ~16 ms; the engine takes three batch contexts, so two stay outstanding while
    // the third is decoded.
    std::vector<Staged> prev, prev2;
    auto finish_all = [&](std::vector<Staged>& v) {
        for (Staged& s : v) finish(s);
        v.clear();
    };
    // Fair share (round 5; the N = 8 ceiling run on the stub backend, tests/test_host_queue.py): a consumer keeps up to
    // three batches going, so near the END of a run — or for a run as short as BASELINE config 4's 2 048 pairs on 8 GPUs —
    // three consumers used to hold nine full batches while five idled (21.6 k instead of 32.7 k pairs/s).  What is left
    // (queued + in flight everywhere) is shared evenly: a consumer that already holds its share delivers its oldest batch
    // before it takes more, and never takes more than what brings it up to the share.
    auto own_inflight = [&] {
        long k = 0;
        for (const Staged& s : prev) k += s.submitted && !s.done;
        for (const Staged& s : prev2) k += s.submitted && !s.done;
        return k;
    };
    auto fair_share = [&] {
        const long total = (long)req_.size() + (shared_ ? shared_->inflight.load() : own_inflight());
        return (total + (long)n_consumers_ - 1) / (long)n_consumers_;
    };
    static const long fair_slack = getenv("TW_FAIR_SLACK") ? atol(getenv("TW_FAIR_SLACK")) : 0;
    for (;;) {
        for (long own = own_inflight(); own > 0 && own >= fair_share() + fair_slack; own = own_inflight()) {
            if (!prev2.empty()) finish_all(prev2);
            else finish_all(prev);
        }
        Request first;
        if (!req_.tryPopNow(first)) {
            finish_all(prev2);  // nothing queued: deliver what is outstanding before blocking
            finish_all(prev);
            read_prof(true);   // idle: nothing of ours is in flight, so the event read's synchronise costs nothing
            publish();
            const long long w0 = steady_ns();
            if (!req_.tryPop(first)) break;
            if (mine.firstJobNs) mine.waitMs += (double)(steady_ns() - w0) * 1e-6;  // blocked on an empty queue
        }
        if (shared_ && shared_->epoch.load() != my_epoch) {
            // Manager::markEpoch(): counters restart here (the manager marks an epoch only while the queue is empty
            // and every response has arrived, so nothing of this consumer is in flight)
            my_epoch = shared_->epoch.load();
            read_prof(false);
            mine.pairs = mine.batches = 0;
            mine.firstJobNs = mine.lastResponseNs = 0;
            mine.waitMs = 0;
            mine.profMs[0] = mine.profMs[1] = 0;
            mine.profLaunches[0] = mine.profLaunches[1] = 0;
        }
        if (!mine.firstJobNs) mine.firstJobNs = steady_ns();
        std::vector<Staged> jobs(1);
        jobs[0].req = std::move(first);
        // a consumer takes at most its share of what is queued, so that a short queue is spread over all GPUs
        // (a greedy grab of `batch` jobs starves the other consumers)
        // (its share of the queue alone, and — above — of everything that is left including what the engines still hold)
        // (round 6: the share of the QUEUE ALONE is no longer a limit of its own.  With it a burst of 2 048 jobs on 8 consumers
        // was handed out as 128, 127, 128, 127, ... and then 1/8 of whatever was left — 112, 98, 85, 49, 42, 25, 21, ... 2, 1,
        // 1, 1: three consumers ended 20 % short of their 256 and one of them worked the tail off in 30 batches of a few
        // pairs.  The fair share of EVERYTHING that is left already spreads a short queue: 10 jobs on 8 consumers are 2 each.)
        const long room = fair_share() + fair_slack - own_inflight();
        const size_t take = std::max<size_t>(1, std::min<size_t>((size_t)batch_, (size_t)std::max(1L, room)));
        static const bool debug_takes = getenv("TW_DEBUG_TAKES") != nullptr;  // diagnostic: who takes what, when
        if (debug_takes)
            fprintf(stderr, "take t=%.3f ms consumer %d queue %zu inflight %ld own %ld share %ld -> take %zu\n", (double)(steady_ns() % 10000000000LL) * 1e-6,
                    id_, req_.size(), shared_ ? shared_->inflight.load() : -1L, own_inflight(), fair_share(), take);
        Request more;
        while (jobs.size() < take && req_.tryPopNow(more)) {
            jobs.emplace_back();
            jobs.back().req = std::move(more);
        }
        if (log_stream())
            for (const Staged& s : jobs)  // src/consumer.cpp:49
                TW_LOGF("consume: %s <-> %s\n", s.req.expect_image.c_str(), s.req.target_image.c_str());
        mine.batches++;
        static const bool batch_time = getenv("TW_DEBUG_BATCHTIME") != nullptr;  // diagnostic: where a batch's host time goes
        const long long bt0 = steady_ns();
        long long bt1 = bt0, bt2 = bt0, bt3 = bt0;
        // decode pool: once the flow runs on the GPU the two imreads of a pair are > 99 % of the wall time
        // (SURVEY §8 f1), so the pairs of a batch are decoded side by side
        {
            std::vector<int> tw(jobs.size(), 0), th(jobs.size(), 0);
            std::vector<std::string> ea(jobs.size()), eb(jobs.size());
            auto parallel_for = [&](size_t ntask, const std::function<void(size_t)>& fn) {
                const size_t nt = std::min(ntask, (size_t)decode_threads_);
                std::atomic<size_t> next{0};
                auto worker = [&] {
                    for (size_t i = next++; i < ntask; i = next++) fn(i);
                };
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nt; t++) pool.emplace_back(worker);
                worker();
                for (std::thread& t : pool) t.join();
            };
            // Phase 1 (pool): read every file.  A PNG the device can finish (8-bit, not interlaced, no palette) is only
            // PARSED here; when every image of the batch is one, the arena is sized from the headers and the pool
            // inflates straight into it (round 4: the intermediate buffers — inflate output, its copy, the copy into the
            // arena, three 2 MB allocations per 1080p image — were a third of a batch's host time).
            struct Opened {
                std::vector<uint8_t> file;
                PngStream st;
                bool read_ok = false, rows_ok = false;
            };
            std::vector<Opened> op(2 * jobs.size());
            parallel_for(2 * jobs.size(), [&](size_t i) {
                const Staged& s = jobs[i >> 1];
                const std::string& path = (i & 1) ? s.req.target_image : s.req.expect_image;
                if (s.req.raw.expect || path.empty()) return;
                Opened& o = op[i];
                o.read_ok = read_file(path, o.file);
                if (o.read_ok && device_png && eng && o.file.size() >= 8 && o.file[0] == 0x89 && o.file[1] == 'P')
                    o.rows_ok = png_parse(o.file, o.st) && o.st.device_rows();
            });
            constexpr size_t kArenaSlotMax = (size_t)64 << 20;  // one 8192 x 8192 image
            constexpr size_t kArenaMax = (size_t)4 << 30;       // per arena, three arenas per consumer
            Arena& ar = arena[arena_idx];  // (its last user was the batch three back: collected below before this one)
            arena_idx = (arena_idx + 1) % 3;
            // the arena is sized from PNG HEADERS, before anything is inflated.  Up to a 4K RGBA screenshot per
            // slot (kArenaFullBatchSlot) it is sized for a full batch at once (no regrowth while a batch fills up); a
            // larger slot gets only what the jobs actually present need, and an arena that an unusual batch blew up is
            // given back as soon as a batch needs less than a quarter of it — one 8192 x 8192 header pair no longer pins
            // 3 x 4 GiB per consumer for the life of the process (worst cases: INTEGRATION.md).
            constexpr size_t kArenaFullBatchSlot = (size_t)40 << 20;
            constexpr size_t kArenaShrinkAbove = (size_t)1 << 30;
            auto ensure_arena = [&](size_t slot) {
                const size_t need = std::min(kArenaMax, slot * 2 * jobs.size());
                if (!eng || !slot) return;
                const bool oversized = ar.base && ar.cap > kArenaShrinkAbove && need <= ar.cap / 4;
                if (need <= ar.cap && !oversized) return;
                if (ar.base) (void)tw_host_free(eng, ar.base);
                ar.base = nullptr;
                ar.cap = 0;
                void* hp = nullptr;
                // a full batch of this size (no regrowth), within the bound — for ordinary slots only
                const size_t full = slot <= kArenaFullBatchSlot ? slot * 2 * (size_t)batch_ : need;
                const size_t want = std::min(kArenaMax, std::max(need, full));
                if (tw_host_alloc(eng, want, &hp) == TW_OK) {
                    ar.base = (uint8_t*)hp;
                    ar.cap = want;
                }
            };
            bool fast = !op.empty();
            size_t fslot = 0;
            for (const Opened& o : op) {
                fast = fast && o.rows_ok;
                fslot = std::max(fslot, (o.st.total + 8 + 255) / 256 * 256);
            }
            fast = fast && fslot <= kArenaSlotMax;
            if (fast) {
                ensure_arena(fslot);
                fast = ar.base && fslot * 2 * jobs.size() <= ar.cap;
            }
            if (fast) {
                // Phase 2 (pool): inflate into the arena slot; every row's filter type is checked as libpng does
                parallel_for(2 * jobs.size(), [&](size_t i) {
                    Opened& o = op[i];
                    uint8_t* dst = ar.base + fslot * i;
                    bool ok = png_inflate(o.st, dst);
                    const size_t rs = (size_t)o.st.w * o.st.ch + 1;
                    for (int y = 0; y < o.st.h && ok; y++) ok = dst[(size_t)y * rs] <= 4;
                    if (!ok) {
                        const Staged& s = jobs[i >> 1];
                        ((i & 1) ? eb : ea)[i >> 1] = "Can't open " + ((i & 1) ? s.req.target_image : s.req.expect_image);
                    }
                    std::vector<uint8_t>().swap(o.file);
                    std::vector<uint8_t>().swap(o.st.idat);
                });
                bt1 = steady_ns();
                for (size_t j = 0; j < jobs.size(); j++) {
                    Staged& s = jobs[j];
                    const PngStream &sa = op[2 * j].st, &sb = op[2 * j + 1].st;
                    // the reference opens the expected image first: its error wins (src/opticalflow.cpp:26-48)
                    if (!ea[j].empty()) { s.err = ea[j]; continue; }
                    if (!eb[j].empty()) { s.err = eb[j]; continue; }
                    s.w = sa.w;
                    s.h = sa.h;
                    s.cha = sa.ch;
                    s.chb = sb.ch;
                    uint8_t* da = ar.base + fslot * (2 * j);
                    uint8_t* db = da + fslot;
                    if (sa.w == sb.w && sa.h == sb.h) {
                        s.pa = da;
                        s.pb = db;
                        s.stride = s.w;
                        continue;
                    }
                    // sizes differ: the <= 5 px reconcile (or the refusal) of prepare(), on copies outside the arena
                    s.a.assign(da, da + sa.total);
                    s.b.assign(db, db + sb.total);
                    prepare(s, sb.w, sb.h, std::string(), std::string());
                }
            } else {
            parallel_for(2 * jobs.size(), [&](size_t i) {
                const size_t j = i >> 1;
                decode_one(jobs[j], (int)(i & 1), &tw[j], &th[j], (i & 1) ? &eb[j] : &ea[j], device_png,
                           op[i].read_ok ? &op[i].file : nullptr);
                std::vector<uint8_t>().swap(op[i].file);
            });
            bt1 = steady_ns();
            parallel_for(jobs.size(), [&](size_t j) { prepare(jobs[j], tw[j], th[j], ea[j], eb[j]); });
            // The arena is sized AFTER prepare(), over the pairs that decoded and reconciled (a header with
            // huge dimensions over a damaged stream must not size it), and it is bounded: a pair larger than
            // kArenaSlotMax, or one the bounded arena has no room for, keeps its pageable buffers (tw_submit_u8 stages
            // those itself — always correct, one memcpy slower).
            size_t slot = 0;
            for (const Staged& s : jobs) {
                if (!s.err.empty() || s.req.raw.expect) continue;
                // (an image is w * h gray bytes or, still filtered, h * (1 + w * ch) bytes)
                const size_t nb = (std::max(s.a.size(), s.b.size()) + 255) / 256 * 256;
                if (nb <= kArenaSlotMax) slot = std::max(slot, nb);
            }
            ensure_arena(slot);
            parallel_for(jobs.size(), [&](size_t j) {
                Staged& s = jobs[j];
                if (!s.err.empty() || s.req.raw.expect || !ar.base || !slot) return;
                if (std::max(s.a.size(), s.b.size()) > slot || (2 * j + 2) * slot > ar.cap) return;  // stays pageable
                uint8_t* da = ar.base + slot * (2 * j);
                uint8_t* db = da + slot;
                memcpy(da, s.a.data(), s.a.size());
                memcpy(db, s.b.data(), s.b.size());
                s.pa = da;
                s.pb = db;
                std::vector<uint8_t>().swap(s.a);
                std::vector<uint8_t>().swap(s.b);
            });
            }  // !fast
        }
        bt2 = steady_ns();
        // one engine batch is homogeneous in size: group equal sizes so that a mixed queue makes few batches
        // (responses are delivered in completion order anyway, like the reference's)
        std::stable_sort(jobs.begin(), jobs.end(), [](const Staged& x, const Staged& y) {
            return x.w != y.w ? x.w < y.w : x.h < y.h;
        });
        for (size_t k = 0; k < jobs.size(); k++) {
            Staged& s = jobs[k];
            if (s.err.empty() && !eng) s.err = eng_err;
            if (!s.err.empty()) continue;
            auto submit = [&]() -> tw_status {
                if (s.cha || s.chb)  // half-decoded PNG(s): the device reconstructs the scanlines
                    return tw_submit_png8(eng, s.pa, s.cha, s.pb, s.chb, s.w, s.h, s.req.span, s.req.threshold, &s.ticket);
                return tw_submit_u8(eng, s.pa, s.pb, s.w, s.h, s.stride, s.req.span, s.req.threshold, &s.ticket);
            };
            tw_status r = submit();
            if (r == TW_E_BUSY) {
                // every batch context of the engine is owed to us (each size change opens one): collect what
                // is outstanding, then this job starts a fresh batch
                finish_all(prev2);
                finish_all(prev);
                for (size_t q = 0; q < k; q++) finish(jobs[q]);
                r = submit();
            }
            if (r == TW_OK) {
                s.submitted = true;
                if (shared_) shared_->inflight++;
            } else s.err = std::string(tw_last_error(eng)[0] ? tw_last_error(eng) : tw_strerror(r));
        }
        if (eng) (void)tw_flush(eng);  // a partly filled batch starts now, not when its first result is asked for
        bt3 = steady_ns();
        const size_t njobs = jobs.size();
        for (Staged& s : jobs)
            if (!s.submitted) finish(s);  // errors do not wait for the GPU
        finish_all(prev2);  // the batch before the previous one: its results have had two batches' time to arrive
        prev2 = std::move(prev);
        prev = std::move(jobs);
        if (batch_time)
            fprintf(stderr, "twhost: consumer %d batch of %zu: decode %.2f ms, prepare + arena %.2f, submit + flush %.2f, "
                            "previous batch's results %.2f\n", id_, njobs, (bt1 - bt0) * 1e-6, (bt2 - bt1) * 1e-6,
                    (bt3 - bt2) * 1e-6, (steady_ns() - bt3) * 1e-6);
    }
    finish_all(prev2);
    finish_all(prev);
    read_prof(true);
    publish();
    for (Arena& ar : arena)
        if (eng && ar.base) (void)tw_host_free(eng, ar.base);
    if (eng) tw_engine_destroy(eng);
    TW_LOGF("finish consumer%d\n", id_);  // src/consumer.cpp:92
}

// ---------------------------------------------------------------------------------------------------
// Manager
// ---------------------------------------------------------------------------------------------------
Manager::~Manager()
{
    stop();
    if (pump_.joinable()) pump_.join();
    for (Consumer* c : consumers_) delete c;
}

void Manager::start(const Parameter& p)
{
    param_ = p;
    running_ = true;
    const int ndev = tw_device_count();
    // numThreads consumers as in src/manager.cpp:55-59, at most `perDevice` of them per GPU (default 1: more
    // consumers than GPUs only time-share a device, the surplus is folded into the per-consumer batch instead;
    // TW_CONSUMERS_PER_DEVICE / Parameter::consumersPerDevice raise it — consumer i runs on device i % ndev)
    int per_dev = p.consumersPerDevice;
    if (per_dev <= 0)
        if (const char* ev = getenv("TW_CONSUMERS_PER_DEVICE")) per_dev = atoi(ev);
    if (per_dev <= 0) per_dev = 1;
    const int n = std::max(1, ndev > 0 ? std::min(p.numThreads, ndev * per_dev) : 1);
    // (a batch also bounds how many images decode side by side: 32 pairs = 64 decode tasks for the pool)
    const int batch = p.batch > 0 ? std::min(256, p.batch) : 32;
    // decode threads per consumer: TW_DECODE_THREADS, else the host cores shared between the consumers
    int dec = 0;
    if (const char* ev = getenv("TW_DECODE_THREADS")) dec = atoi(ev);
    if (dec <= 0) dec = std::max(1, std::min(16, (int)std::thread::hardware_concurrency() / n));
    shared_.profile = p.profileKernels;
    shared_.stats.assign((size_t)n, ConsumerStats());
    for (int i = 0; i < n; i++) {
        consumers_.push_back(new Consumer(i, requestQueue_, responseQueue_, p.optParam, batch, dec, n, &shared_));
        consumers_.back()->start();
    }
    pump_ = std::thread([this] { work(); });
}

void Manager::waitReady()
{
    std::unique_lock<std::mutex> lk(shared_.m);
    shared_.cv.wait(lk, [&] {
        for (const ConsumerStats& s : shared_.stats)
            if (!s.ready) return false;
        return true;
    });
}

std::vector<ConsumerStats> Manager::consumerStats()
{
    std::lock_guard<std::mutex> lk(shared_.m);
    return shared_.stats;
}

void Manager::markEpoch()
{
    shared_.epoch++;
    std::lock_guard<std::mutex> lk(report_m_);
    pump_stats_ = PumpStats();
    pump_sum_us_ = 0;
}

PumpStats Manager::pumpStats()
{
    std::lock_guard<std::mutex> lk(report_m_);
    PumpStats p = pump_stats_;
    p.meanUs = p.delivered ? pump_sum_us_ / (double)p.delivered : 0;
    return p;
}

int Manager::request(const std::string& expect_image, const std::string& target_image)
{
    Request r;
    r.expect_image = expect_image;
    r.target_image = target_image;
    r.span = param_.span;
    r.threshold = param_.threshold;
    TW_LOGF("request: %s <-> %s\n", expect_image.c_str(), target_image.c_str());  // src/manager.cpp:74
    {
        std::lock_guard<std::mutex> lk(report_m_);
        report_.requestCount++;
    }
    requestQueue_.push(std::move(r));
    return 0;
}

int Manager::requestRaw(const std::string& expect_name, const std::string& target_name, const RawPair& raw)
{
    Request r;
    r.expect_image = expect_name;
    r.target_image = target_name;
    r.span = param_.span;
    r.threshold = param_.threshold;
    r.raw = raw;
    {
        std::lock_guard<std::mutex> lk(report_m_);
        report_.requestCount++;
    }
    requestQueue_.push(std::move(r));
    return 0;
}

void Manager::stop()
{
    bool expected = false;
    if (!stopped_.compare_exchange_strong(expected, true)) return;
    running_ = false;
    responseQueue_.stop();
}

void Manager::work()
{
    Response res;
    while (responseQueue_.tryPop(res)) {  // src/manager.cpp:80-90 + notify() :102-116
        const long long t_pop = steady_ns();
        const bool is_err = res.status == "ERROR";
        {
            std::lock_guard<std::mutex> lk(report_m_);
            if (is_err) report_.errorCount++;
            else report_.dataCount++;
            if (res.pushedNs) {
                const double us = (double)(t_pop - res.pushedNs) * 1e-3;
                pump_stats_.delivered++;
                pump_sum_us_ += us;
                if (us > pump_stats_.maxUs) pump_stats_.maxUs = us;
            }
        }
        if (is_err) {
            if (obs_.onError) obs_.onError(res.reason);
        } else {
            if (obs_.onNext) obs_.onNext(res);
        }
        const double cb_ms = (double)(steady_ns() - t_pop) * 1e-6;
        std::lock_guard<std::mutex> lk(report_m_);
        pump_stats_.busyMs += cb_ms;
    }
    requestQueue_.stop();  // src/manager.cpp:93-97
    for (Consumer* c : consumers_) c->join();
    Report rep;
    {
        std::lock_guard<std::mutex> lk(report_m_);
        rep = report_;
    }
    if (obs_.onCompleted) obs_.onCompleted(rep);
}

}  // namespace twhost
