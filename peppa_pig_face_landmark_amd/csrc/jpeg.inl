// pf_jpeg_info / pf_decode_jpeg: JPEG file -> packed BGR frame in device memory (frame ingest; the
// reference does cv2.imread(path) on the host and hands the array to FaceAna.run, demo.py:76).  Host: marker parsing and the
// Huffman-coded scan(s) -> 16-bit coefficient blocks in page-locked memory (ITU T.81 F.2.2; jdhuff.c is the model for the
// bit reader: byte stuffing, restart intervals, missing data decodes as zeros).  Device: k_jpeg.h.
// Supported: 8-bit baseline / extended-sequential Huffman (SOF0 / SOF1), greyscale or YCbCr with luma sampling 1x1, 2x1 or 2x2
// over 1x1 chroma (4:4:4, 4:2:2, 4:2:0), interleaved or one scan per component, restart markers.  Anything else -- progressive,
// arithmetic coding, 12-bit, CMYK / Adobe RGB, other sampling grids -- is refused with a message, never decoded approximately.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JpegHuff {
    bool present = false;
    unsigned char bits[17] = {0};
    unsigned char vals[256] = {0};
    int maxcode[18];      // largest code of each length, -1 if none (maxcode[17] = sentinel)
    int valoff[17];       // vals index of the first code of a length minus that code
    unsigned short look[512];   // 9-bit prefix -> (length << 8) | symbol, 0 = longer code
    short fast_ac[512];         // AC tables: 9-bit prefix holds code AND magnitude bits -> (value << 8) | (run << 4) | total bits, else 0
    bool build() {
        int code = 0, k = 0;
        for (int i = 0; i < 512; ++i) look[i] = 0;
        for (int l = 1; l <= 16; ++l) {
            valoff[l] = k - code;
            if (bits[l]) {
                for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                    // an over-subscribed table (more codes of length l than 2^l code points are left) must be refused BEFORE
                    // its codes index the lookahead table: code << (9 - l) would run past look[] into the neighbouring members
                    if (k >= 256 || code >= (1 << l)) return false;
                    if (l <= 9) {
                        const int lo = code << (9 - l), n = 1 << (9 - l);
                        for (int j = 0; j < n; ++j) look[lo + j] = (unsigned short)((l << 8) | vals[k]);
                    }
                }
                maxcode[l] = code - 1;
            } else {
                maxcode[l] = -1;
            }
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int i = 0; i < 512; ++i) {
            fast_ac[i] = 0;
            const int lk = look[i];
            if (!lk) continue;
            const int len = lk >> 8, rs = lk & 0xFF, run = rs >> 4, sz = rs & 15;
            if (sz == 0 || len + sz > 9) continue;
            int v = (i >> (9 - len - sz)) & ((1 << sz) - 1);          // the magnitude bits that follow the code
            if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
            if (v >= -128 && v <= 127) fast_ac[i] = (short)((v * 256) | (run << 4) | (len + sz));
        }
        return true;
    }
};

struct JpegComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;       // blocks per row / column of the plane (padded to whole MCUs)
    int dw = 0, dh = 0;       // downsampled width / height that carry image data
    int block0 = 0;           // first block in the coefficient buffer
    int pred = 0;
};

struct JpegHeader {
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0, total_blocks = 0;
    JpegComp c[3];
    unsigned short q[4][64];      // natural order
    bool qpresent[4] = {false, false, false, false};
    JpegHuff dc[4], ac[4];
    bool have_sof = false, adobe = false;
    int adobe_transform = -1;
};

struct JpegBits {
    const unsigned char* p;
    const unsigned char* end;
    unsigned long long acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 48) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0) ++p;                 // stuffed zero
                    else { --p; hit_marker = true; b = 0; }      // a marker: the segment ends, feed zeros (jdhuff.c does the same)
                }
            } else {
                hit_marker = true;
            }
            acc = (acc << 8) | (unsigned)b;
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)((acc >> (n - k)) & ((1u << k) - 1)); }
    int peek_nofill(int k) const { return (int)((acc >> (n - k)) & ((1u << k) - 1)); }   // caller keeps n >= 32
    void skip(int k) { n -= k; }
    int get(int k) { if (k == 0) return 0; const int v = peek(k); n -= k; return v; }
    void align_reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int jpeg_extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

inline int jpeg_huff_decode(JpegBits& br, const JpegHuff& t) {
    const int look = t.look[br.peek(9)];
    if (look) { br.skip(look >> 8); return look & 0xFF; }
    int code = br.peek(9), l = 9;
    br.skip(9);
    for (;;) {
        code = (code << 1) | br.get(1);
        ++l;
        if (l > 16) return 0;                    // garbage in the stream: decode as a zero-length symbol
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) return t.vals[(code + t.valoff[l]) & 0xFF];
    }
}

inline unsigned jpeg_be16(const unsigned char* p) { return ((unsigned)p[0] << 8) | p[1]; }

// Parses the header up to (not including) the first SOS; *sos_at = offset of that marker's 0xFF.  Returns an error text or nullptr.
const char* jpeg_parse_header(const unsigned char* d, size_t n, JpegHeader& hd, size_t* sos_at) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return "not a JPEG stream (no SOI)";
    size_t pos = 2;
    for (;;) {
        while (pos < n && d[pos] != 0xFF) ++pos;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) return "truncated JPEG header";
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return "JPEG ends before any scan";
        if (pos + 2 > n) return "truncated JPEG header";
        const size_t len = jpeg_be16(d + pos);
        if (len < 2 || pos + len > n) return "bad JPEG segment length";
        const unsigned char* s = d + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xDA) { *sos_at = pos - 2; break; }
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return "only 8-bit JPEG samples are supported";
            hd.H = (int)jpeg_be16(s + 1); hd.W = (int)jpeg_be16(s + 3); hd.ncomp = s[5];
            if (hd.H < 1 || hd.W < 1) return "JPEG with zero dimensions (DNL) is not supported";
            if (hd.ncomp != 1 && hd.ncomp != 3) return "only greyscale and 3-component (YCbCr) JPEGs are supported";
            if (sl < (size_t)6 + 3 * hd.ncomp) return "truncated SOF";
            for (int i = 0; i < hd.ncomp; ++i) {
                JpegComp& c = hd.c[i];
                c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                if (c.tq > 3) return "bad quantisation table index";
            }
            hd.have_sof = true;
        } else if (m == 0xC2) {
            return "progressive JPEG is not supported (re-encode as baseline)";
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return "lossless / hierarchical / arithmetic-coded JPEG is not supported";
        } else if (m == 0xCC) {
            return "arithmetic-coded JPEG is not supported";
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                ++o;
                if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > sl) return "bad DQT segment";
                for (int k = 0; k < 64; ++k) {
                    hd.q[tq][kZigzag[k]] = pq ? (unsigned short)jpeg_be16(s + o + 2 * k) : s[o + k];
                }
                o += pq ? 128 : 64;
                hd.qpresent[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                if (o + 17 > sl) return "bad DHT segment";
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return "bad DHT table id";
                JpegHuff& t = tc ? hd.ac[th] : hd.dc[th];
                int cnt = 0;
                t.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[o + l]; cnt += t.bits[l]; }
                o += 17;
                if (cnt > 256 || o + cnt > sl) return "bad DHT segment";
                for (int k = 0; k < cnt; ++k) t.vals[k] = s[o + k];
                o += cnt;
                if (!t.build()) return "bad Huffman table";
                t.present = true;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return "bad DRI segment";
            hd.restart = (int)jpeg_be16(s);
        } else if (m == 0xEE) {
            if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') { hd.adobe = true; hd.adobe_transform = s[11]; }
        }
        pos += len;
    }
    if (!hd.have_sof) return "no SOF0 / SOF1 frame header before the scan";
    if (hd.ncomp == 3 && hd.adobe && hd.adobe_transform == 0) return "Adobe RGB JPEG (no YCbCr transform) is not supported";
    if (hd.ncomp == 1) { hd.c[0].h = hd.c[0].v = 1; }                 // a single component's sampling factors are irrelevant
    else {
        if (hd.c[1].h != 1 || hd.c[1].v != 1 || hd.c[2].h != 1 || hd.c[2].v != 1) return "chroma sampling factors other than 1x1 are not supported";
        const int h = hd.c[0].h, v = hd.c[0].v;
        if (!((h == 1 && v == 1) || (h == 2 && v == 1) || (h == 2 && v == 2))) return "only 4:4:4, 4:2:2 and 4:2:0 sampling are supported";
    }
    hd.hmax = hd.c[0].h; hd.vmax = hd.c[0].v;
    hd.mcux = (hd.W + 8 * hd.hmax - 1) / (8 * hd.hmax);
    hd.mcuy = (hd.H + 8 * hd.vmax - 1) / (8 * hd.vmax);
    int blocks = 0;
    for (int i = 0; i < hd.ncomp; ++i) {
        JpegComp& c = hd.c[i];
        c.bw = hd.mcux * c.h; c.bh = hd.mcuy * c.v;
        c.dw = (hd.W * c.h + hd.hmax - 1) / hd.hmax; c.dh = (hd.H * c.v + hd.vmax - 1) / hd.vmax;
        c.block0 = blocks;
        blocks += c.bw * c.bh;
        if (!hd.qpresent[c.tq]) return "quantisation table missing";
    }
    hd.total_blocks = blocks;
    return nullptr;
}

// One block -> a packed record at rec: [length][coefficients in zigzag order up to the last non-zero one].  Returns the record's
// size in 16-bit units.
inline int jpeg_decode_block(JpegBits& br, const JpegHuff& dct, const JpegHuff& act, int& pred, short* rec) {
    const int s = jpeg_huff_decode(br, dct);
    const int diff = s ? jpeg_extend(br.get(s > 15 ? 15 : s), s > 15 ? 15 : s) : 0;
    pred += diff;
    rec[1] = (short)pred;
    int last = 0;
    for (int k = 1; k < 64;) {
        if (br.n < 32) br.fill();                // one refill covers a code (<= 16 bits) and its magnitude (<= 15)
        const int fa = act.fast_ac[br.peek_nofill(9)];
        int r, val;
        if (fa) {                                // code and magnitude inside the 9-bit window: one lookup
            br.skip(fa & 15);
            r = (fa >> 4) & 15;
            val = fa >> 8;
        } else {
            const int rs = jpeg_huff_decode(br, act);
            const int sz = rs & 15;
            r = rs >> 4;
            if (sz == 0) {
                if (r != 15) break;              // EOB
                k += 16;
                continue;
            }
            val = jpeg_extend(br.peek_nofill(sz), sz);
            br.skip(sz);
        }
        k += r;
        if (k > 63) break;
        for (int z = last + 1; z < k; ++z) rec[1 + z] = 0;
        rec[1 + k] = (short)val;
        last = k;
        ++k;
    }
    rec[0] = (short)(last + 1);
    return last + 2;
}

// All scans -> packed block records (tab[block] = record start in 16-bit units, 0xFFFFFFFF where no scan reached the block).  Returns an error text or nullptr.
const char* jpeg_decode_scans(const unsigned char* d, size_t n, size_t pos, JpegHeader& hd, unsigned* tab, short* data, size_t* used) {
    size_t cur = 0;
    for (int b = 0; b < hd.total_blocks; ++b) tab[b] = 0xFFFFFFFFu;
    bool seen[3] = {false, false, false};
    for (;;) {
        // at a marker
        while (pos < n && d[pos] != 0xFF) ++pos;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD9) break;
        if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (pos + 2 > n) break;
        const size_t len = jpeg_be16(d + pos);
        if (len < 2 || pos + len > n) return "bad JPEG segment length";
        const unsigned char* s = d + pos + 2;
        if (m == 0xC4 || m == 0xDB || m == 0xDD) {               // tables may change between scans
            // re-use the header parser's table code on this one segment (wrapped as SOI, the segment, an empty SOS)
            unsigned char hdr[4] = {0xFF, 0xD8, 0xFF, (unsigned char)m};
            std::vector<unsigned char> buf(hdr, hdr + 4);
            buf.insert(buf.end(), d + pos, d + pos + len);
            const unsigned char tail[4] = {0xFF, 0xDA, 0x00, 0x02};
            buf.insert(buf.end(), tail, tail + 4);
            size_t dummy = 0;
            JpegHeader h2 = hd;
            if (const char* e = jpeg_parse_header(buf.data(), buf.size(), h2, &dummy)) return e;
            for (int i = 0; i < 4; ++i) { hd.dc[i] = h2.dc[i]; hd.ac[i] = h2.ac[i]; hd.qpresent[i] = h2.qpresent[i]; for (int k = 0; k < 64; ++k) hd.q[i][k] = h2.q[i][k]; }
            hd.restart = h2.restart;
            pos += len;
            continue;
        }
        if (m != 0xDA) { pos += len; continue; }
        const size_t sl = len - 2;
        if (sl < 1) return "bad SOS segment";
        const int ns = s[0];
        if (ns < 1 || ns > hd.ncomp || sl < (size_t)1 + 2 * ns + 3) return "bad SOS segment";
        int ci[3];
        for (int i = 0; i < ns; ++i) {
            int found = -1;
            for (int j = 0; j < hd.ncomp; ++j) if (hd.c[j].id == s[1 + 2 * i]) found = j;
            if (found < 0) return "SOS names an unknown component";
            ci[i] = found;
            hd.c[found].td = s[2 + 2 * i] >> 4; hd.c[found].ta = s[2 + 2 * i] & 15;
            if (hd.c[found].td > 3 || hd.c[found].ta > 3 || !hd.dc[hd.c[found].td].present || !hd.ac[hd.c[found].ta].present) return "Huffman table missing";
            if (seen[found]) return "a component appears in more than one scan";
            seen[found] = true;
        }
        if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return "spectral selection / successive approximation need a progressive decoder";
        pos += len;
        JpegBits br{d + pos, d + n};
        for (int j = 0; j < hd.ncomp; ++j) hd.c[j].pred = 0;
        int mcus_x, mcus_y;
        if (ns == 1) { const JpegComp& c = hd.c[ci[0]]; mcus_x = (c.dw + 7) / 8; mcus_y = (c.dh + 7) / 8; }
        else { mcus_x = hd.mcux; mcus_y = hd.mcuy; }
        int until_restart = hd.restart, next_rst = 0;
        for (int my = 0; my < mcus_y; ++my) {
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (hd.restart && until_restart == 0) {
                    // byte-align, find RSTn
                    const unsigned char* q = br.p;
                    // bytes already pulled into the accumulator beyond the marker are zeros fed after hit_marker; step to the marker
                    while (q + 1 < d + n && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) break;      // some other marker: stop looking
                        ++q;
                    }
                    if (q + 1 < d + n && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
                    br.p = q;
                    br.align_reset();
                    next_rst = (next_rst + 1) & 7;
                    until_restart = hd.restart;
                    for (int j = 0; j < hd.ncomp; ++j) hd.c[j].pred = 0;
                }
                if (ns == 1) {
                    JpegComp& c = hd.c[ci[0]];
                    const size_t b = (size_t)c.block0 + (size_t)my * c.bw + mx;
                    tab[b] = (unsigned)cur;
                    cur += jpeg_decode_block(br, hd.dc[c.td], hd.ac[c.ta], c.pred, data + cur);
                } else {
                    for (int i = 0; i < ns; ++i) {
                        JpegComp& c = hd.c[ci[i]];
                        for (int v = 0; v < c.v; ++v)
                            for (int h = 0; h < c.h; ++h) {
                                const size_t b = (size_t)c.block0 + (size_t)(my * c.v + v) * c.bw + (mx * c.h + h);
                                tab[b] = (unsigned)cur;
                                cur += jpeg_decode_block(br, hd.dc[c.td], hd.ac[c.ta], c.pred, data + cur);
                            }
                    }
                }
                if (hd.restart) --until_restart;
            }
        }
        // continue after the entropy-coded segment: br.p sits at (or shortly before) the next marker
        pos = (size_t)(br.p - d);
    }
    for (int j = 0; j < hd.ncomp; ++j) if (!seen[j]) return "a component has no scan";
    *used = cur;
    return nullptr;
}

// Device entropy decoding applies when the file has ONE interleaved scan in frame component order and table ids 0 / 1 (with or
// without restart markers): copies the scan, the interval offsets and the lookup tables into the frame's pinned region.  Returns false (file untouched)
// when the file does not qualify or its markers do not add up -- the host decoder then takes it.
bool jpeg_stage_for_device(const unsigned char* d, size_t n, size_t sos, JpegHeader& hd, unsigned char* region, size_t region_cap,
                           JpegFrameDesc* desc, size_t* used_bytes) {
    if (sos + 4 > n || d[sos] != 0xFF || d[sos + 1] != 0xDA) return false;
    const size_t len = jpeg_be16(d + sos + 2);
    if (len < 2 || sos + 2 + len > n) return false;
    const unsigned char* s = d + sos + 4;
    const size_t sl = len - 2;
    if (sl < 1 || s[0] != hd.ncomp || sl < (size_t)1 + 2 * hd.ncomp + 3) return false;
    for (int i = 0; i < hd.ncomp; ++i) {
        if (s[1 + 2 * i] != hd.c[i].id) return false;
        hd.c[i].td = s[2 + 2 * i] >> 4; hd.c[i].ta = s[2 + 2 * i] & 15;
        if (hd.c[i].td > 1 || hd.c[i].ta > 1 || !hd.dc[hd.c[i].td].present || !hd.ac[hd.c[i].ta].present) return false;
    }
    if (s[1 + 2 * hd.ncomp] != 0 || s[2 + 2 * hd.ncomp] != 63 || s[3 + 2 * hd.ncomp] != 0) return false;
    const size_t begin = sos + 2 + len;
    const int total_mcus = hd.mcux * hd.mcuy;
    const int want = hd.restart > 0 ? (total_mcus + hd.restart - 1) / hd.restart : 1;     // no restart markers: one unbroken stream
    if (n - begin + 64 + (size_t)want * 4 + sizeof(JpegGpuTables) + 64 > region_cap) return false;
    // The scan is copied WITHOUT its byte stuffing and without the RSTn markers (offsets in the copied stream): the device threads
    // then refill their bit buffers four bytes at a time, branch-free -- with stuffing in place the 64 lanes of a wave each looped
    // over single bytes and the wave paid the longest loop on every symbol.
    std::vector<unsigned> offs;
    offs.reserve((size_t)want);
    offs.push_back(0);
    size_t q = begin, out = 0;
    while (q < n) {
        const unsigned char* ff = static_cast<const unsigned char*>(memchr(d + q, 0xFF, n - q));
        const size_t run = ff ? (size_t)(ff - (d + q)) : n - q;
        memcpy(region + out, d + q, run);
        out += run; q += run;
        if (!ff || q + 1 >= n) break;
        const int m = d[q + 1];
        if (m == 0x00) { region[out++] = 0xFF; q += 2; continue; }
        if (m == 0xFF) { q += 1; continue; }
        if (m >= 0xD0 && m <= 0xD7) { offs.push_back((unsigned)out); q += 2; continue; }
        if (m != 0xD9) return false;               // more scans / tables follow: not the single-scan case
        break;
    }
    if ((int)offs.size() != want) return false;
    const size_t scan_len = out;
    memset(region + out, 0, 32);                   // the bit readers run a few bytes past the end of the last interval
    const size_t offs_off = (scan_len + 32 + 15) / 16 * 16;
    const size_t tables_off = (offs_off + offs.size() * 4 + 15) / 16 * 16;
    if (tables_off + sizeof(JpegGpuTables) > region_cap) return false;
    memcpy(region + offs_off, offs.data(), offs.size() * 4);
    JpegGpuTables* t = reinterpret_cast<JpegGpuTables*>(region + tables_off);
    for (int i = 0; i < 4; ++i) {
        const JpegHuff& src = i < 2 ? hd.dc[i] : hd.ac[i - 2];
        memcpy(t->look[i], src.look, sizeof(src.look));
        memcpy(t->fast_ac[i], src.fast_ac, sizeof(src.fast_ac));
        memcpy(t->maxcode[i], src.maxcode, sizeof(src.maxcode));
        memcpy(t->valoff[i], src.valoff, sizeof(src.valoff));
        memcpy(t->vals[i], src.vals, sizeof(src.vals));
        t->limit[i][0] = 0;
        for (int l = 1; l <= 16; ++l)                  // lengths without codes keep the bound of the shorter ones
            t->limit[i][l] = src.maxcode[l] >= 0 ? (src.maxcode[l] + 1) << (16 - l) : t->limit[i][l - 1];
    }
    if (hd.restart <= 0) {                         // the sub-sequence decoder's limits: checked BEFORE *desc is touched, so a refusal
        int bpm = 0;                               // never leaves a half-filled descriptor that reads as a device frame
        for (int i = 0; i < hd.ncomp; ++i) bpm += hd.c[i].h * hd.c[i].v;
        if (hd.ncomp > 3 || bpm > 10) return false;
    }
    desc->scan_off = 0; desc->scan_len = (unsigned)scan_len;
    desc->offs_off = (unsigned)offs_off; desc->n_intervals = (unsigned)offs.size();
    desc->tables_off = (unsigned)tables_off; desc->restart = (unsigned)std::max(hd.restart, 0); desc->pad = 0;
    if (hd.restart <= 0) {                         // decoded as sub-sequences of PF_JPEG_SUBSEQ_BITS bits (k_jpeg.h jpeg_sync_kernel)
        desc->n_intervals = (unsigned)std::max<size_t>(1, (scan_len * 8 + PF_JPEG_SUBSEQ_BITS - 1) / PF_JPEG_SUBSEQ_BITS);
    }
    desc->tdta = 0;
    for (int i = 0; i < hd.ncomp; ++i) desc->tdta |= ((unsigned)hd.c[i].td << i) | ((unsigned)hd.c[i].ta << (4 + i));
    *used_bytes = tables_off + sizeof(JpegGpuTables);
    return true;
}

}  // namespace

void JpegState::release() {
    for (Slot& sl : slot) {
        if (sl.h_pack) (void)hipHostFree(sl.h_pack);
        if (sl.d_pack) (void)hipFree(sl.d_pack);
        if (sl.d_bgr) (void)hipFree(sl.d_bgr);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.consumed) (void)hipEventDestroy(sl.consumed);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (d_coef) (void)hipFree(d_coef);
    if (d_planes) (void)hipFree(d_planes);
    if (d_quant) (void)hipFree(d_quant);
    if (d_desc) (void)hipFree(d_desc);
    if (d_sub) (void)hipFree(d_sub);
    *this = JpegState{};
}

extern "C" {

int pf_jpeg_info(const uint8_t* jpeg, size_t bytes, int* height, int* width, int* components, int* subsampling) {
    if (!jpeg) return 1;
    JpegHeader hd;
    size_t sos = 0;
    if (jpeg_parse_header(jpeg, bytes, hd, &sos)) return 1;
    if (height) *height = hd.H;
    if (width) *width = hd.W;
    if (components) *components = hd.ncomp;
    if (subsampling) *subsampling = hd.ncomp == 1 ? 0 : (hd.hmax == 1 ? 444 : (hd.vmax == 1 ? 422 : 420));
    return 0;
}

// n equally shaped JPEGs -> [n][H][W][3] in device memory; the entropy decoding of the files runs on `threads` host threads
static int jpeg_decode_batch(pf_handle* h, int n, const uint8_t* const* jpegs, const size_t* sizes, int threads,
                             int* height, int* width, const uint8_t** d_bgr, uint8_t* bgr_host, bool final_sync, bool force_host = false) {
    if (n < 1 || !jpegs || !sizes) PF_FAIL(h, "pf_decode_jpeg: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    std::vector<JpegHeader> hds((size_t)n);
    std::vector<size_t> sos((size_t)n, 0);
    for (int f = 0; f < n; ++f) {
        if (!jpegs[f]) PF_FAIL(h, "pf_decode_jpeg: null input");
        if (const char* e = jpeg_parse_header(jpegs[f], sizes[f], hds[f], &sos[f])) PF_FAIL(h, "pf_decode_jpeg: %s", e);
        if (f && (hds[f].W != hds[0].W || hds[f].H != hds[0].H || hds[f].ncomp != hds[0].ncomp || hds[f].hmax != hds[0].hmax ||
                  hds[f].vmax != hds[0].vmax))
            PF_FAIL(h, "pf_decode_jpeg_batch: image %d differs in size or sampling from image 0", f);
    }
    const JpegHeader& hd = hds[0];
    JpegState& s = h->jpeg;
    s.cur ^= 1;
    JpegState::Slot& sl = s.slot[s.cur];
    if (!s.copy_stream) PF_HIP(h, hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking));
    if (!sl.uploaded) {
        PF_HIP(h, hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        PF_HIP(h, hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));
    }
    if (sl.in_flight) PF_HIP(h, hipEventSynchronize(sl.consumed));   // two decodes ago: long done; the slot's buffers are free
    sl.in_flight = false;
    const size_t coef_bytes = (size_t)hd.total_blocks * 64 * sizeof(short) * n;
    const size_t plane_bytes = (size_t)hd.total_blocks * 64;
    if (s.coef_cap < coef_bytes || s.planes_cap < plane_bytes * n || s.quant_cap < (size_t)n) {
        PF_HIP(h, hipStreamSynchronize(h->stream));                  // intermediates are shared by both slots: drain before growing
        if (s.d_coef) (void)hipFree(s.d_coef);
        if (s.d_planes) (void)hipFree(s.d_planes);
        if (s.d_quant) (void)hipFree(s.d_quant);
        s.d_coef = nullptr; s.d_planes = nullptr; s.d_quant = nullptr; s.coef_cap = s.planes_cap = s.quant_cap = 0;
        PF_HIP(h, hipMalloc((void**)&s.d_coef, coef_bytes));
        PF_HIP(h, hipMalloc((void**)&s.d_planes, plane_bytes * n));
        PF_HIP(h, hipMalloc((void**)&s.d_quant, (size_t)n * 3 * 64 * sizeof(unsigned short)));
        s.coef_cap = coef_bytes; s.planes_cap = plane_bytes * n; s.quant_cap = (size_t)n;
    }
    // packed records: worst case 65 values per block (every block is written by exactly one scan: decode_scans refuses a component that appears twice)
    const size_t frame_pack = ((size_t)hd.total_blocks * (4 + 65 * sizeof(short)) + 255) / 256 * 256;
    if (sl.pack_cap < frame_pack * n) {
        PF_HIP(h, hipStreamSynchronize(h->stream));
        if (sl.h_pack) (void)hipHostFree(sl.h_pack);
        if (sl.d_pack) (void)hipFree(sl.d_pack);
        sl.h_pack = nullptr; sl.d_pack = nullptr; sl.pack_cap = 0;
        PF_HIP(h, hipHostMalloc((void**)&sl.h_pack, frame_pack * n, hipHostMallocPortable));
        PF_HIP(h, hipMalloc((void**)&sl.d_pack, frame_pack * n));
        sl.pack_cap = frame_pack * n;
    }
    const size_t out_bytes = (size_t)hd.H * hd.W * 3 * n;
    if (sl.bgr_cap < out_bytes) {
        PF_HIP(h, hipStreamSynchronize(h->stream));
        if (sl.d_bgr) (void)hipFree(sl.d_bgr);
        sl.d_bgr = nullptr; sl.bgr_cap = 0;
        h->alloc_epoch++;                          // a captured graph may hold the old frame pointer
        PF_HIP(h, hipMalloc((void**)&sl.d_bgr, out_bytes));
        sl.bgr_cap = out_bytes;
    }
    // ---- entropy decoding: one file per task ------------------------------------------------------------------------------------
    std::vector<const char*> errs((size_t)n, nullptr);
    std::vector<size_t> used((size_t)n, 0);          // bytes of each frame's region that cross PCIe
    std::vector<JpegFrameDesc> descs((size_t)n);
    // Where the Huffman stream is decoded.  A restart interval is one GPU thread, and a GPU thread is slow (a 120-MCU interval of a
    // 1080p file takes ~7 ms): the device wins when the batch offers thousands of intervals (96 files x 68: 11.8 k files/s against
    // 3.6-5.2 k on 16-96 host threads), a single file is faster on one host core (4 ms).  PF_OPT_JPEG_ENTROPY overrides.
    // A file without restart markers is cut into 1024-bit sub-sequences that synchronise themselves (k_jpeg.h jpeg_sync_kernel): a
    // 1080p file offers ~3 500 of them, so that path goes to the device for any batch size.
    bool device_entropy = hd.restart <= 0;
    if (hd.restart > 0) {
        const long long intervals = (long long)n * ((hd.mcux * hd.mcuy + hd.restart - 1) / hd.restart);
        device_entropy = intervals >= 4096;
    }
    if (h->jpeg_entropy != 0) device_entropy = h->jpeg_entropy == 2;          // PF_OPT_JPEG_ENTROPY: 1 host, 2 device
    if (force_host) device_entropy = false;
    int rounds = PF_JPEG_SYNC_ROUNDS;
    if (h->jpeg_rounds > 0) rounds = std::min(PF_JPEG_SYNC_ROUNDS, h->jpeg_rounds);                                    // PF_OPT_JPEG_SYNC_ROUNDS
    std::vector<std::atomic<int>> done((size_t)n);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    // workers are pure CPU (a HIP call from a fresh thread pays the runtime's per-thread set-up under its global lock: 60 ms
    // stalls with three lanes decoding at once); the calling thread uploads each file's records as soon as that file is done
    auto work = [&](int first, int step) {
        for (int f = first; f < n; f += step) {
            unsigned char* region = sl.h_pack + (size_t)f * frame_pack;
            descs[f] = JpegFrameDesc{};
            if (device_entropy && jpeg_stage_for_device(jpegs[f], sizes[f], sos[f], hds[f], region, frame_pack, &descs[f], &used[f])) {
                // nothing else to do on the host: the scan is decoded by jpeg_huffman_kernel
            } else {
                size_t values = 0;
                errs[f] = jpeg_decode_scans(jpegs[f], sizes[f], sos[f], hds[f], reinterpret_cast<unsigned*>(region),
                                            reinterpret_cast<short*>(region + (size_t)hd.total_blocks * 4), &values);
                used[f] = (size_t)hd.total_blocks * 4 + values * sizeof(short);
            }
            done[f].store(1, std::memory_order_release);
        }
    };
    const auto t_start = std::chrono::steady_clock::now();
    const int T = std::max(1, std::min(threads, n));
    std::vector<std::thread> pool;
    if (T > 1)
        for (int t = 0; t < T; ++t) pool.emplace_back(work, t, T);
    else
        work(0, 1);
    hipError_t copy_rc = hipSuccess;
    for (int f = 0; f < n; ++f) {        // only what the decoder wrote crosses PCIe: the block table and the records
        while (!done[f].load(std::memory_order_acquire)) std::this_thread::yield();
        if (errs[f] || copy_rc != hipSuccess) continue;
        copy_rc = hipMemcpyAsync(sl.d_pack + (size_t)f * frame_pack, sl.h_pack + (size_t)f * frame_pack, used[f],
                                 hipMemcpyHostToDevice, s.copy_stream);
    }
    for (auto& th : pool) th.join();
    for (int f = 0; f < n; ++f)
        if (errs[f]) PF_FAIL(h, "pf_decode_jpeg: image %d: %s", f, errs[f]);
    PF_HIP(h, copy_rc);
    const auto t_entropy = std::chrono::steady_clock::now();
    PF_HIP(h, hipEventRecord(sl.uploaded, s.copy_stream));
    PF_HIP(h, hipStreamWaitEvent(h->stream, sl.uploaded, 0));        // the engine's stream expands them once they are all there
    int on_device = 0, max_iv = 0;
    for (int f = 0; f < n; ++f) { on_device += descs[f].n_intervals ? 1 : 0; max_iv = std::max(max_iv, (int)descs[f].n_intervals); }
    if (s.desc_cap < (size_t)n) {
        if (s.d_desc) (void)hipFree(s.d_desc);
        s.d_desc = nullptr; s.desc_cap = 0;
        PF_HIP(h, hipMalloc((void**)&s.d_desc, (size_t)n * sizeof(JpegFrameDesc)));
        s.desc_cap = (size_t)n;
    }
    PF_HIP(h, hipMemcpyAsync(s.d_desc, descs.data(), (size_t)n * sizeof(JpegFrameDesc), hipMemcpyHostToDevice, h->stream));
    if (on_device < n) {
        JpegUnpackArgs ua{};
        ua.pack = sl.d_pack; ua.frame_pack_bytes = frame_pack; ua.coef = s.d_coef; ua.blocks = hd.total_blocks; ua.desc = s.d_desc;
        ProfScope ps(h, "jpeg_unpack");
        PF_LAUNCH(jpeg_unpack_kernel, dim3((unsigned)pf_div_up(hd.total_blocks, 64), (unsigned)n), dim3(64), h->stream, ua);
    }
    int on_sub = 0, max_sub = 0;
    for (int f = 0; f < n; ++f)
        if (descs[f].n_intervals && descs[f].restart == 0) { ++on_sub; max_sub = std::max(max_sub, (int)descs[f].n_intervals); }
    if (on_sub) {
        const size_t words = (size_t)n * max_sub * 5 + 16;
        if (s.sub_cap < words) {
            PF_HIP(h, hipStreamSynchronize(h->stream));
            if (s.d_sub) (void)hipFree(s.d_sub);
            s.d_sub = nullptr; s.sub_cap = 0;
            PF_HIP(h, hipMalloc((void**)&s.d_sub, words * sizeof(unsigned)));
            s.sub_cap = words;
        }
        JpegSubseqArgs qa{};
        qa.pack = sl.d_pack; qa.frame_pack_bytes = frame_pack; qa.desc = s.d_desc; qa.coef = s.d_coef; qa.blocks = hd.total_blocks;
        qa.ncomp = hd.ncomp; qa.mcux = hd.mcux; qa.total_mcus = hd.mcux * hd.mcuy; qa.bpm = 0;
        for (int c = 0; c < hd.ncomp; ++c) {
            qa.ch[c] = hd.c[c].h; qa.cv[c] = hd.c[c].v; qa.cblock0[c] = hd.c[c].block0; qa.cbw[c] = hd.c[c].bw;
            qa.bpm += hd.c[c].h * hd.c[c].v;
        }
        const size_t per = (size_t)n * max_sub;
        qa.exit_p = s.d_sub; qa.exit_s = s.d_sub + per; qa.nblk = s.d_sub + 2 * per; qa.blk0 = s.d_sub + 3 * per; qa.stamp = s.d_sub + 4 * per;
        qa.changed = s.d_sub + 5 * per;
        qa.max_sub = max_sub;
        PF_HIP(h, hipMemsetAsync(qa.changed, 0, 16 * sizeof(unsigned), h->stream));
        // (the write pass stores only what it decodes: zero the blocks first)
        if (on_sub == n) {
            PF_HIP(h, hipMemsetAsync(s.d_coef, 0, coef_bytes, h->stream));
        } else {
            for (int f = 0; f < n; ++f)
                if (descs[f].n_intervals && descs[f].restart == 0)
                    PF_HIP(h, hipMemsetAsync(s.d_coef + (size_t)f * hd.total_blocks * 64, 0, (size_t)hd.total_blocks * 64 * sizeof(short), h->stream));
        }
        const dim3 sg((unsigned)pf_div_up(max_sub, 64), (unsigned)n);
        {
            ProfScope ps(h, "jpeg_subseq");              // first pass + synchronisation rounds + block-ordinal scan
            qa.round = 0;
            PF_LAUNCH(jpeg_sync_kernel<0>, sg, dim3(64), h->stream, qa);
            for (int r = 1; r <= rounds; ++r) {
                qa.round = r;
                PF_LAUNCH(jpeg_sync_kernel<1>, sg, dim3(64), h->stream, qa);
            }
            PF_LAUNCH(jpeg_subseq_scan_kernel, dim3((unsigned)n), dim3(256), h->stream, qa);
        }
        {
            ProfScope ps(h, "jpeg_subseq_write");        // write pass + DC prefix + verdict
            qa.round = -1;
            PF_LAUNCH(jpeg_sync_kernel<2>, sg, dim3(64), h->stream, qa);
            PF_LAUNCH(jpeg_dc_prefix_kernel, dim3((unsigned)hd.ncomp, (unsigned)n), dim3(256), h->stream, qa);
            PF_LAUNCH(jpeg_subseq_verdict_kernel, dim3(1), dim3(1), h->stream, qa.changed + rounds, h->h_status);
        }
    }
    if (on_device > on_sub) {
        // (every block of such a frame is written whole by the kernel: an interleaved scan covers the MCU-padded planes)
        JpegHuffArgs ha{};
        ha.pack = sl.d_pack; ha.frame_pack_bytes = frame_pack; ha.desc = s.d_desc; ha.coef = s.d_coef; ha.blocks = hd.total_blocks;
        ha.ncomp = hd.ncomp; ha.mcux = hd.mcux; ha.total_mcus = hd.mcux * hd.mcuy;
        for (int c = 0; c < hd.ncomp; ++c) {
            ha.ch[c] = hd.c[c].h; ha.cv[c] = hd.c[c].v; ha.cblock0[c] = hd.c[c].block0; ha.cbw[c] = hd.c[c].bw;
        }
        ProfScope ps(h, "jpeg_huffman");
        PF_LAUNCH(jpeg_huffman_kernel, dim3((unsigned)pf_div_up(max_iv, 64), (unsigned)n), dim3(64), h->stream, ha);
    }
    PF_HIP(h, hipEventRecord(sl.consumed, h->stream));
    sl.in_flight = true;
    std::vector<unsigned short> qt((size_t)n * 3 * 64, 0);
    for (int f = 0; f < n; ++f)
        for (int c = 0; c < hd.ncomp; ++c)
            for (int k = 0; k < 64; ++k) qt[((size_t)f * 3 + c) * 64 + k] = hds[f].q[hds[f].c[c].tq][k];
    PF_HIP(h, hipMemcpyAsync(s.d_quant, qt.data(), qt.size() * sizeof(unsigned short), hipMemcpyHostToDevice, h->stream));
    JpegIdctArgs ia{};
    ia.coef = s.d_coef; ia.ncomp = hd.ncomp; ia.quant = s.d_quant; ia.frame_blocks = (size_t)hd.total_blocks;
    for (int c = 0; c < hd.ncomp; ++c) {
        ia.block0[c] = hd.c[c].block0; ia.bw[c] = hd.c[c].bw;
        ia.plane[c] = s.d_planes + (size_t)hd.c[c].block0 * 64;
    }
    ia.block0[hd.ncomp] = hd.total_blocks;
    {
        ProfScope ps(h, "jpeg_idct");
        PF_LAUNCH(jpeg_idct_kernel, dim3((unsigned)pf_div_up(hd.total_blocks, 64), (unsigned)n), dim3(64), h->stream, ia);
    }
    JpegColorArgs ca{};
    ca.y = ia.plane[0]; ca.ys = hd.c[0].bw * 8;
    ca.W = hd.W; ca.H = hd.H; ca.out = sl.d_bgr; ca.frame_plane_bytes = plane_bytes;
    if (hd.ncomp == 1) {
        ca.mode = 0;
    } else {
        ca.cb = ia.plane[1]; ca.cr = ia.plane[2]; ca.cs = hd.c[1].bw * 8;
        ca.cw = hd.c[1].dw; ca.ch = hd.c[1].dh;
        ca.mode = hd.hmax == 1 ? 1 : (hd.vmax == 1 ? 2 : 3);
        if (ca.mode >= 2 && hd.c[1].dw <= 2) ca.mode += 4;       // jdsample.c: the triangle filter needs more than two columns
    }
    {
        ProfScope ps(h, "jpeg_color");
        if ((hd.W & 3) == 0 && (reinterpret_cast<uintptr_t>(sl.d_bgr) & 3) == 0)
            PF_LAUNCH(jpeg_color4_kernel, dim3((unsigned)(((long long)(hd.W / 4) * hd.H + 255) / 256), (unsigned)n), dim3(256), h->stream, ca);
        else
            PF_LAUNCH(jpeg_color_kernel, dim3((unsigned)(((long long)hd.W * hd.H + 255) / 256), (unsigned)n), dim3(256), h->stream, ca);
    }
    if (bgr_host) PF_HIP(h, hipMemcpyAsync(bgr_host, sl.d_bgr, out_bytes, hipMemcpyDeviceToHost, h->stream));
    if (final_sync || bgr_host) {
        PF_HIP(h, hipStreamSynchronize(h->stream));
        if (h->h_status && h->h_status[0] == 3 && !force_host) {     // the sub-sequence decoder did not settle: the host decodes this batch
            h->h_status[0] = 0;
            return jpeg_decode_batch(h, n, jpegs, sizes, threads, height, width, d_bgr, bgr_host, final_sync, true);
        }
        if (check_numerics(h)) return 1;
    }
    if (PF_ABLATE != 0 && getenv("PEPPA_JPEG_TIMING")) {      // tool build only (tools/jpeg_profile.py)
        const auto t_end = std::chrono::steady_clock::now();
        size_t up = 0;
        for (int f = 0; f < n; ++f) up += used[f];
        fprintf(stderr, "[peppa-hip] jpeg batch of %d: entropy decode %.3f ms on %d thread(s), upload (%.2f MB) + device stages %.3f ms\n", n,
                std::chrono::duration<double, std::milli>(t_entropy - t_start).count(), T, up / 1e6,
                std::chrono::duration<double, std::milli>(t_end - t_entropy).count());
    }
    if (height) *height = hd.H;
    if (width) *width = hd.W;
    if (d_bgr) *d_bgr = sl.d_bgr;
    return 0;
}

int pf_decode_jpeg(pf_handle* h, const uint8_t* jpeg, size_t bytes, int* height, int* width, const uint8_t** d_bgr, uint8_t* bgr_host) {
    if (!h) return 1;
    return jpeg_decode_batch(h, 1, &jpeg, &bytes, 1, height, width, d_bgr, bgr_host, true);
}

int pf_decode_jpeg_batch(pf_handle* h, int n, const uint8_t* const* jpegs, const size_t* sizes, int threads, int* height, int* width,
                         const uint8_t** d_frames) {
    if (!h) return 1;
    return jpeg_decode_batch(h, n, jpegs, sizes, threads, height, width, d_frames, nullptr, false);
}

}  // extern "C"
