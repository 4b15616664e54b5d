// jpeg_gray.cpp — JPEG -> 8-bit gray, what cv::imread(path, 0) hands the flow (src/opticalflow.cpp:37,44 of the
// reference; OpenCV 2.4.9 grfmt_jpeg.cpp asks libjpeg for out_color_space = JCS_GRAYSCALE, which for a
// YCbCr or gray file is the luma plane: chroma is never converted or upsampled).
//
// Written from the JPEG standard (ITU-T T.81): baseline / extended sequential (SOF0, SOF1) and progressive
// (SOF2) Huffman streams, 8-bit precision, restart intervals, any sampling as long as luma has the largest
// factors.  The inverse DCT is the "slow integer" one (jidctint: Loeffler-Ligtenberg-Moschytz, 13-bit
// constants, 2 extra bits after the column pass) that every libjpeg lineage uses by default, so the bytes are
// the ones libjpeg produces.  Not handled (-> false, the job answers "Can't open <path>" like an imread
// failure): arithmetic coding, lossless, 12-bit, RGB/CMYK/YCCK colour spaces, luma subsampled below chroma.
//
// Pin: tests/test_node_addon.py compares decodeGray() with the gray decode of the reference's JPEG fixture and
// of JPEGs written by PIL (libjpeg-turbo: baseline/progressive, 4:4:4/4:2:2/4:2:0, gray, restart markers).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "twhost.h"

namespace twhost {
namespace {

const uint8_t kZigzag[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
    6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
    39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a corrupt run can step past 63: park those writes on the last coefficient
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool present = false;
    uint8_t vals[256];
    int maxcode[18];   // largest code of each length, -1 if none
    int valoff[17];    // index of the first value of a length minus its first code
    uint8_t look_n[512], look_v[512];  // 9-bit lookahead: code length (0 = longer) and symbol

    bool build(const uint8_t* bits /*[1..16]*/, const uint8_t* v, int nv)
    {
        memset(vals, 0, sizeof(vals));
        memcpy(vals, v, (size_t)nv);
        memset(look_n, 0, sizeof(look_n));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valoff[l] = k - code;
            if (bits[l]) {
                for (int i = 0; i < bits[l]; i++, k++, code++) {
                    if (l <= 9) {
                        const int first = code << (9 - l);
                        for (int f = 0; f < (1 << (9 - l)); f++) {
                            look_n[first + f] = (uint8_t)l;
                            look_v[first + f] = vals[k];
                        }
                    }
                }
                maxcode[l] = code - 1;
                if (code > (1 << l)) return false;
            } else {
                maxcode[l] = -1;
            }
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool marker = false;  // ran into a marker: feed zero bits, as libjpeg does

    void fill()
    {
        while (n <= 24) {
            uint32_t c = 0;
            if (!marker && p < end) {
                c = *p++;
                if (c == 0xFF) {
                    if (p < end && *p == 0) {
                        p++;
                    } else {
                        p--;
                        marker = true;
                        c = 0;
                    }
                }
            }
            acc |= c << (24 - n);
            n += 8;
        }
    }
    int get(int k)
    {
        if (k == 0) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k;
        n -= k;
        return v;
    }
    int bit() { return get(1); }
    int sym(const Huff& h)
    {
        if (n < 16) fill();
        const int top = (int)(acc >> 23);
        int l = h.look_n[top];
        if (l) {
            acc <<= l;
            n -= l;
            return h.look_v[top];
        }
        int code = (int)(acc >> 22);  // 10 bits
        for (l = 10; l <= 16; l++) {
            if (code <= h.maxcode[l]) break;
            code = (int)(acc >> (32 - l - 1));
        }
        if (l > 16) {
            acc <<= 16;
            n -= 16;
            return 0;  // bad code: libjpeg warns and returns 0
        }
        acc <<= l;
        n -= l;
        return h.vals[(code + h.valoff[l]) & 255];
    }
    void reset_to(const uint8_t* q)
    {
        p = q;
        acc = 0;
        n = 0;
        marker = false;
    }
};

inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

struct Comp {
    int id = 0, H = 1, V = 1, tq = 0;
    int bw = 0, bh = 0;      // blocks stored (padded to whole MCUs)
    int cw = 0, ch = 0;      // blocks a non-interleaved scan of this component covers
    std::vector<int16_t> coef;
    int dc_tbl = 0, ac_tbl = 0, pred = 0;
};

// jidctint: dequantised coefficient block (natural order) -> 64 samples
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride)
{
    constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                      F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int64_t ws[64];
    auto pass = [&](const int64_t* v /*8 inputs*/, int shift, int64_t* o /*8 outputs*/) {
        int64_t z2 = v[2], z3 = v[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847);
        int64_t tmp3 = z1 + z2 * F0_765;
        z2 = v[0];
        z3 = v[4];
        int64_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = v[7];
        tmp1 = v[5];
        tmp2 = v[3];
        tmp3 = v[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298;
        tmp1 *= F2_053;
        tmp2 *= F3_072;
        tmp3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        const int64_t rnd = (int64_t)1 << (shift - 1);
        o[0] = (tmp10 + tmp3 + rnd) >> shift;
        o[7] = (tmp10 - tmp3 + rnd) >> shift;
        o[1] = (tmp11 + tmp2 + rnd) >> shift;
        o[6] = (tmp11 - tmp2 + rnd) >> shift;
        o[2] = (tmp12 + tmp1 + rnd) >> shift;
        o[5] = (tmp12 - tmp1 + rnd) >> shift;
        o[3] = (tmp13 + tmp0 + rnd) >> shift;
        o[4] = (tmp13 - tmp0 + rnd) >> shift;
    };
    for (int c = 0; c < 8; c++) {
        int64_t v[8], o[8];
        for (int r = 0; r < 8; r++) v[r] = (int64_t)in[8 * r + c] * q[8 * r + c];
        pass(v, 13 - 2, o);
        for (int r = 0; r < 8; r++) ws[8 * r + c] = o[r];
    }
    for (int r = 0; r < 8; r++) {
        int64_t o[8];
        pass(ws + 8 * r, 13 + 2 + 3, o);
        for (int c = 0; c < 8; c++) {
            // libjpeg's range_limit table indexed with (x & 1023): clamp(x + 128) for any sane x
            const int v = (int)(o[c] & 1023);
            out[r * out_stride + c] = (uint8_t)(v < 128 ? v + 128 : (v < 512 ? 255 : (v < 896 ? 0 : v - 896)));
        }
    }
}

struct Decoder {
    const uint8_t* d;
    size_t n;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, have_frame = false;
    Comp comp[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart_interval = 0;
    int adobe_transform = -1;

    static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

    bool parse_dqt(const uint8_t* p, int len)
    {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) return false;
            const int need = 1 + 64 * (pq + 1);
            if (len < need) return false;
            for (int i = 0; i < 64; i++) qt[tq][kZigzag[i]] = (uint16_t)(pq ? be16(p + 1 + 2 * i) : p[1 + i]);
            qt_present[tq] = true;
            p += need;
            len -= need;
        }
        return true;
    }
    bool parse_dht(const uint8_t* p, int len)
    {
        while (len > 0) {
            if (len < 17) return false;
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) return false;
            uint8_t bits[17];
            bits[0] = 0;
            int nv = 0;
            for (int i = 1; i <= 16; i++) nv += (bits[i] = p[i]);
            if (nv > 256 || len < 17 + nv) return false;
            if (!(tc ? ac[th] : dc[th]).build(bits, p + 17, nv)) return false;
            p += 17 + nv;
            len -= 17 + nv;
        }
        return true;
    }
    bool parse_sof(const uint8_t* p, int len)
    {
        if (len < 6 || have_frame) return false;
        if (p[0] != 8) return false;  // 12-bit: not an 8-bit gray source
        H = be16(p + 1);
        W = be16(p + 3);
        ncomp = p[5];
        if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3) || len < 6 + 3 * ncomp) return false;
        if ((long long)W * H > (1LL << 27)) return false;  // coefficient storage for a whole frame is kept in memory
        for (int i = 0; i < ncomp; i++) {
            Comp& c = comp[i];
            c.id = p[6 + 3 * i];
            c.H = p[7 + 3 * i] >> 4;
            c.V = p[7 + 3 * i] & 15;
            c.tq = p[8 + 3 * i];
            if (c.H < 1 || c.H > 4 || c.V < 1 || c.V > 4 || c.tq > 3) return false;
            hmax = std::max(hmax, c.H);
            vmax = std::max(vmax, c.V);
        }
        if (ncomp == 1) comp[0].H = comp[0].V = hmax = vmax = 1;  // a single component is never interleaved
        if (comp[0].H != hmax || comp[0].V != vmax) return false;  // luma would need upsampling
        mcux = (W + 8 * hmax - 1) / (8 * hmax);
        mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Comp& c = comp[i];
            c.bw = mcux * c.H;
            c.bh = mcuy * c.V;
            const int pw = (W * c.H + hmax - 1) / hmax, ph = (H * c.V + vmax - 1) / vmax;
            c.cw = (pw + 7) / 8;
            c.ch = (ph + 7) / 8;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        have_frame = true;
        return true;
    }

    // ---- entropy-coded segment of one scan ----
    struct Scan {
        int n = 0;
        int ci[4];
        int Ss = 0, Se = 63, Ah = 0, Al = 0;
    };

    void block_seq(Bits& b, Comp& c, int16_t* blk)
    {
        const int s = b.sym(dc[c.dc_tbl]);
        const int diff = s ? extend(b.get(s & 15), s & 15) : 0;
        c.pred += diff;
        blk[0] = (int16_t)c.pred;
        const Huff& h = ac[c.ac_tbl];
        for (int k = 1; k < 64; k++) {
            const int rs = b.sym(h), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                blk[kZigzag[k]] = (int16_t)extend(b.get(sz), sz);
            } else {
                if (r != 15) break;
                k += 15;
            }
        }
    }
    void block_dc_first(Bits& b, Comp& c, int16_t* blk, int Al)
    {
        const int s = b.sym(dc[c.dc_tbl]);
        const int diff = s ? extend(b.get(s & 15), s & 15) : 0;
        c.pred += diff;
        blk[0] = (int16_t)(c.pred * (1 << Al));
    }
    void block_ac_first(Bits& b, Comp& c, int16_t* blk, const Scan& sc, int& eobrun)
    {
        if (eobrun > 0) {
            eobrun--;
            return;
        }
        const Huff& h = ac[c.ac_tbl];
        for (int k = sc.Ss; k <= sc.Se; k++) {
            const int rs = b.sym(h), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                blk[kZigzag[k]] = (int16_t)(extend(b.get(sz), sz) * (1 << sc.Al));
            } else {
                if (r == 15) {
                    k += 15;
                } else {
                    eobrun = 1 << r;
                    if (r) eobrun += b.get(r);
                    eobrun--;
                    break;
                }
            }
        }
    }
    void block_ac_refine(Bits& b, Comp& c, int16_t* blk, const Scan& sc, int& eobrun)
    {
        const int p1 = 1 << sc.Al, m1 = -(1 << sc.Al);
        const Huff& h = ac[c.ac_tbl];
        int k = sc.Ss;
        auto refine = [&](int16_t& v) {
            if (b.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            for (; k <= sc.Se; k++) {
                const int rs = b.sym(h);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = b.bit() ? p1 : m1;  // a new coefficient is always +-1 at this bit position
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += b.get(r);
                    break;
                }
                // step over r still-zero coefficients, refining the non-zero ones passed on the way
                do {
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0) {
                        refine(v);
                    } else if (--r < 0) {
                        break;
                    }
                    k++;
                } while (k <= sc.Se);
                if (s) blk[kZigzag[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= sc.Se; k++) {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0) refine(v);
            }
            eobrun--;
        }
    }

    // returns the position of the marker that ends the scan
    const uint8_t* decode_scan(const uint8_t* p, const Scan& sc)
    {
        Bits b;
        b.p = p;
        b.end = d + n;
        for (int i = 0; i < sc.n; i++) comp[sc.ci[i]].pred = 0;
        int eobrun = 0;
        const bool inter = sc.n > 1;
        Comp& c0 = comp[sc.ci[0]];
        const int mx = inter ? mcux : c0.cw, my = inter ? mcuy : c0.ch;
        int togo = restart_interval, next_rst = 0;
        auto one = [&](Comp& c, int16_t* blk) {
            if (!progressive) block_seq(b, c, blk);
            else if (sc.Ss == 0) {
                if (sc.Ah == 0) block_dc_first(b, c, blk, sc.Al);
                else if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << sc.Al));
            } else if (sc.Ah == 0) block_ac_first(b, c, blk, sc, eobrun);
            else block_ac_refine(b, c, blk, sc, eobrun);
        };
        for (int y = 0; y < my; y++) {
            for (int x = 0; x < mx; x++) {
                if (restart_interval && togo == 0) {
                    // byte-align, find RSTn, reset the predictors
                    const uint8_t* q = b.p;
                    if (!b.marker) {
                        // unread whole bytes still sit in the accumulator: the marker follows the consumed data
                        q = b.p - (b.n / 8);
                    }
                    while (q + 1 < d + n && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) break;  // some other marker: give up resync
                        q++;
                    }
                    if (q + 1 < d + n && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
                    next_rst = (next_rst + 1) & 7;
                    b.reset_to(q);
                    for (int i = 0; i < sc.n; i++) comp[sc.ci[i]].pred = 0;
                    eobrun = 0;
                    togo = restart_interval;
                }
                if (inter) {
                    for (int i = 0; i < sc.n; i++) {
                        Comp& c = comp[sc.ci[i]];
                        for (int v = 0; v < c.V; v++)
                            for (int hh = 0; hh < c.H; hh++)
                                one(c, &c.coef[((size_t)(y * c.V + v) * c.bw + (x * c.H + hh)) * 64]);
                    }
                } else {
                    one(c0, &c0.coef[((size_t)y * c0.bw + x) * 64]);
                }
                if (restart_interval) togo--;
            }
        }
        // the next marker: at b.p if the reader stopped on it, else scan forward from the consumed position
        const uint8_t* q = b.marker ? b.p : b.p - (b.n / 8);
        if (q < p) q = p;
        while (q + 1 < d + n && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        return q;
    }

    bool run(std::vector<uint8_t>& img, int& w, int& h)
    {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
        const uint8_t* p = d + 2;
        const uint8_t* end = d + n;
        bool seen_scan = false, eoi = false;
        while (p + 4 <= end && !eoi) {
            if (p[0] != 0xFF) {
                p++;
                continue;
            }
            const int m = p[1];
            if (m == 0xFF) {
                p++;
                continue;
            }
            if (m == 0xD9) {
                eoi = true;
                break;
            }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) {
                p += 2;
                continue;
            }
            const int len = be16(p + 2);
            if (len < 2 || p + 2 + len > end) break;
            const uint8_t* body = p + 4;
            const int blen = len - 2;
            bool ok = true;
            if (m == 0xDB) ok = parse_dqt(body, blen);
            else if (m == 0xC4) ok = parse_dht(body, blen);
            else if (m == 0xC0 || m == 0xC1) ok = parse_sof(body, blen);
            else if (m == 0xC2) {
                progressive = true;
                ok = parse_sof(body, blen);
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) ok = false;  // lossless / arithmetic
            else if (m == 0xCC) ok = false;
            else if (m == 0xDD) {
                if (blen < 2) ok = false;
                else restart_interval = be16(body);
            } else if (m == 0xEE && blen >= 12 && memcmp(body, "Adobe", 5) == 0) adobe_transform = body[11];
            else if (m == 0xDA) {
                if (!have_frame || blen < 1) return false;
                Scan sc;
                sc.n = body[0];
                if (sc.n < 1 || sc.n > ncomp || blen < 1 + 2 * sc.n + 3) return false;
                for (int i = 0; i < sc.n; i++) {
                    int ci = -1;
                    for (int j = 0; j < ncomp; j++)
                        if (comp[j].id == body[1 + 2 * i]) ci = j;
                    if (ci < 0) return false;
                    sc.ci[i] = ci;
                    comp[ci].dc_tbl = body[2 + 2 * i] >> 4;
                    comp[ci].ac_tbl = body[2 + 2 * i] & 15;
                    if (comp[ci].dc_tbl > 3 || comp[ci].ac_tbl > 3) return false;
                }
                sc.Ss = body[1 + 2 * sc.n];
                sc.Se = body[2 + 2 * sc.n];
                sc.Ah = body[3 + 2 * sc.n] >> 4;
                sc.Al = body[3 + 2 * sc.n] & 15;
                if (!progressive) {
                    sc.Ss = 0;
                    sc.Se = 63;
                    sc.Ah = sc.Al = 0;
                } else {
                    if (sc.Ss > sc.Se || sc.Se > 63 || sc.Al > 13) return false;
                    if (sc.Ss == 0 && sc.Se != 0) return false;
                    if (sc.Ss > 0 && sc.n != 1) return false;
                }
                // tables the scan needs
                for (int i = 0; i < sc.n; i++) {
                    const Comp& c = comp[sc.ci[i]];
                    const bool need_dc = !progressive || (sc.Ss == 0 && sc.Ah == 0);
                    const bool need_ac = !progressive || sc.Ss > 0;
                    if (need_dc && !dc[c.dc_tbl].present) return false;
                    if (need_ac && !ac[c.ac_tbl].present) return false;
                }
                p = decode_scan(p + 2 + len, sc);
                seen_scan = true;
                continue;
            }
            if (!ok) return false;
            p += 2 + len;
        }
        if (!have_frame || !seen_scan) return false;
        // RGB-coded JPEG (Adobe transform 0, or component ids 'R','G','B'): gray would need a colour conversion
        if (ncomp == 3 && adobe_transform == 0) return false;
        if (ncomp == 3 && adobe_transform < 0 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') return false;
        Comp& Y = comp[0];
        if (!qt_present[Y.tq]) return false;
        // inverse DCT of the luma blocks, cropped to the image
        const int pw = Y.bw * 8;
        std::vector<uint8_t> plane((size_t)pw * Y.bh * 8);
        for (int by = 0; by < Y.bh; by++)
            for (int bx = 0; bx < Y.bw; bx++)
                idct_islow(&Y.coef[((size_t)by * Y.bw + bx) * 64], qt[Y.tq], &plane[(size_t)by * 8 * pw + bx * 8], pw);
        img.resize((size_t)W * H);
        for (int y = 0; y < H; y++) memcpy(&img[(size_t)y * W], &plane[(size_t)y * pw], (size_t)W);
        w = W;
        h = H;
        return true;
    }
};

}  // namespace

bool decode_jpeg_gray(const uint8_t* data, size_t n, std::vector<uint8_t>& img, int& w, int& h)
{
    Decoder dec;
    dec.d = data;
    dec.n = n;
    return dec.run(img, w, h);
}

}  // namespace twhost
