// The part of the device JPEG decoder that reads the file: header parser, Huffman table builder, byte un-stuffing into the
// staging buffer (see jpeg.hip for the decoder, jpeg_plan.h for why this is a translation unit of its own).
#include "jpeg_plan.h"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace mvlm_jpeg;

namespace {

const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffHost {
    bool have = false;
    uint8_t counts[16];
    uint8_t vals[256];
    int n = 0;
};


constexpr int MAX_SIDE = 16384;

inline int rd16(const uint8_t* p) { return (int(p[0]) << 8) | p[1]; }

void build_slot(JpegDev& d, int slot, const HuffHost& h) {
    std::memset(d.lut[slot], 0, sizeof(d.lut[slot]));
    std::memcpy(d.vals[slot], h.vals, 256);
    for (int i = 0; i < 8; ++i) {
        d.limit[slot][i] = 0x10000u;
        d.valptr[slot][i] = 0;
    }
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        const int n = h.counts[len - 1];
        if (len >= LUT_BITS) d.valptr[slot][len - LUT_BITS] = uint8_t(k);
        for (int i = 0; i < n; ++i, ++code, ++k) {
            if (len <= LUT_BITS) {
                const uint32_t first = code << (LUT_BITS - len);
                for (uint32_t f = 0; f < (1u << (LUT_BITS - len)); ++f) d.lut[slot][first + f] = uint16_t((len << 8) | h.vals[k]);
            }
        }
        if (len >= LUT_BITS) d.limit[slot][len - LUT_BITS] = code << (16 - len);
        code <<= 1;
    }
}

void no_codes_slot(JpegDev& d, int slot) {  // a slot no component uses: every window is invalid
    std::memset(d.lut[slot], 0, sizeof(d.lut[slot]));
    for (int i = 0; i < 8; ++i) {
        d.limit[slot][i] = 0;
        d.valptr[slot][i] = 0;
    }
}

}  // namespace

// 0: a JPEG the device decoder takes; 2: not taken (why says which rule); never touches the GPU
int mvlm_jpeg_plan_impl(const uint8_t* data, size_t n, MvlmJpegPlan& plan, std::string& why) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) {
        why = "not a JPEG stream";
        return 2;
    }
    if (n >= (size_t(1) << 30)) {
        why = "file too large";
        return 2;
    }
    JpegDev& d = plan.dev;
    std::memset(&d, 0, sizeof(d));
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    HuffHost huff[2][4];
    int comp_id[3] = {0, 0, 0}, comp_h[3] = {1, 1, 1}, comp_v[3] = {1, 1, 1}, comp_tq[3] = {0, 0, 0}, comp_td[3], comp_ta[3];
    bool have_frame = false, jfif = false;
    int adobe = -1;
    size_t p = 2;
    for (;;) {
        while (p < n && data[p] != 0xFF) ++p;
        while (p < n && data[p] == 0xFF) ++p;
        if (p >= n) {
            why = "no scan";
            return 2;
        }
        const int m = data[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9 || p + 2 > n) {
            why = "no scan";
            return 2;
        }
        const size_t len = size_t(rd16(data + p));
        if (len < 2 || p + len > n) {
            why = "truncated segment";
            return 2;
        }
        const uint8_t* seg = data + p + 2;
        const size_t sl = len - 2;
        p += len;
        if (m == 0xDB) {
            size_t q = 0;
            while (q < sl) {
                const int pq = seg[q] >> 4, tq = seg[q] & 15;
                ++q;
                if (tq > 3 || q + (pq ? 128 : 64) > sl) {
                    why = "bad DQT";
                    return 2;
                }
                for (int i = 0; i < 64; ++i) {
                    quant[tq][kNatural[i]] = pq ? uint16_t(rd16(seg + q + 2 * i)) : seg[q + i];
                    if (quant[tq][kNatural[i]] > 255) {  // (the 32-bit inverse DCT here is exact for 8-bit tables only)
                        why = "quantisation table with 16-bit entries";
                        return 2;
                    }
                }
                q += pq ? 128 : 64;
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t q = 0;
            while (q < sl) {
                if (q + 17 > sl) {
                    why = "bad DHT";
                    return 2;
                }
                const int tc = seg[q] >> 4, th = seg[q] & 15;
                int cnt = 0;
                for (int i = 0; i < 16; ++i) cnt += seg[q + 1 + i];
                if (tc > 1 || th > 3 || cnt > 256 || q + 17 + size_t(cnt) > sl) {
                    why = "bad DHT";
                    return 2;
                }
                HuffHost& h = huff[tc][th];
                h.have = true;
                h.n = cnt;
                std::memcpy(h.counts, seg + q + 1, 16);
                std::memset(h.vals, 0, 256);
                std::memcpy(h.vals, seg + q + 17, size_t(cnt));
                // the codes must fit their lengths (Kraft): a table that over-subscribes a length is corrupt
                long code = 0;
                for (int l = 1; l <= 16; ++l) {
                    code += h.counts[l - 1];
                    if (code > (1l << l)) {
                        why = "bad DHT";
                        return 2;
                    }
                    code <<= 1;
                }
                q += 17 + size_t(cnt);
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || seg[0] != 8) {
                why = "sample precision other than 8 bit";
                return 2;
            }
            d.height = rd16(seg + 1);
            d.width = rd16(seg + 3);
            d.ncomp = seg[5];
            if ((d.ncomp != 1 && d.ncomp != 3) || sl < size_t(6 + 3 * d.ncomp)) {
                why = "component count other than 1 or 3";
                return 2;
            }
            for (int i = 0; i < d.ncomp; ++i) {
                comp_id[i] = seg[6 + 3 * i];
                comp_h[i] = seg[7 + 3 * i] >> 4;
                comp_v[i] = seg[7 + 3 * i] & 15;
                comp_tq[i] = seg[8 + 3 * i];
            }
            have_frame = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            why = "progressive / lossless / arithmetic JPEG";
            return 2;
        } else if (m == 0xDD) {
            if (sl < 2) {
                why = "bad DRI";
                return 2;
            }
            plan.ri = rd16(seg);
        } else if (m == 0xE0 && sl >= 5 && std::memcmp(seg, "JFIF\0", 5) == 0) {
            jfif = true;
        } else if (m == 0xEE && sl >= 12 && std::memcmp(seg, "Adobe", 5) == 0) {
            adobe = seg[11];
        } else if (m == 0xDA) {
            if (!have_frame || sl < 1 || seg[0] != d.ncomp || sl < size_t(4 + 2 * d.ncomp)) {
                why = "a scan that does not hold every component";
                return 2;
            }
            for (int i = 0; i < d.ncomp; ++i) {
                if (seg[1 + 2 * i] != comp_id[i]) {
                    why = "scan components out of frame order";
                    return 2;
                }
                comp_td[i] = seg[2 + 2 * i] >> 4;
                comp_ta[i] = seg[2 + 2 * i] & 15;
            }
            if (seg[1 + 2 * d.ncomp] != 0 || seg[2 + 2 * d.ncomp] != 63 || seg[3 + 2 * d.ncomp] != 0) {
                why = "spectral selection / successive approximation in a sequential scan";
                return 2;
            }
            break;
        }
    }
    if (d.width <= 0 || d.height <= 0) {
        why = "empty frame";
        return 2;
    }
    // The header is untrusted and everything below is sized from it: refuse an implausible frame BEFORE any arithmetic on it
    // (a side of at most MAX_SIDE keeps every product below in int range; the reference's own scanner textures are
    // 3546 x 2282) - the caller's host decoder then applies its own limits.
    if (d.width > MAX_SIDE || d.height > MAX_SIDE) {
        why = "frame larger than 16384 pixels a side";
        return 2;
    }
    if (d.ncomp == 3) {
        bool ycc;  // jdapimin.c default_decompress_parms
        if (jfif)
            ycc = true;
        else if (adobe >= 0)
            ycc = adobe != 0;
        else
            ycc = !(comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B');
        if (!ycc) {
            why = "three components that are not YCbCr";
            return 2;
        }
        const bool luma_ok = (comp_h[0] == 1 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 2);
        if (!luma_ok || comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) {
            why = "sampling factors other than 1x1 / 2x1 / 2x2 luma over 1x1 chroma";
            return 2;
        }
    } else {
        comp_h[0] = comp_v[0] = 1;
    }
    d.hmax = comp_h[0];
    d.vmax = comp_v[0];
    d.mcus_x = (d.width + 8 * d.hmax - 1) / (8 * d.hmax);
    d.mcus_y = (d.height + 8 * d.vmax - 1) / (8 * d.vmax);
    d.bpm = 0;
    size_t off = 0;
    for (int c = 0; c < d.ncomp; ++c) {
        if (comp_tq[c] > 3 || !have_q[comp_tq[c]] || comp_td[c] > 3 || comp_ta[c] > 3 || !huff[0][comp_td[c]].have || !huff[1][comp_ta[c]].have) {
            why = "a table the scan names is missing";
            return 2;
        }
        std::memcpy(d.quant[c], quant[comp_tq[c]], sizeof(d.quant[c]));
        build_slot(d, c, huff[0][comp_td[c]]);
        build_slot(d, 3 + c, huff[1][comp_ta[c]]);
        for (int by = 0; by < comp_v[c]; ++by)
            for (int bx = 0; bx < comp_h[c]; ++bx) {
                d.blk_comp[d.bpm] = c;
                d.blk_bx[d.bpm] = bx;
                d.blk_by[d.bpm] = by;
                ++d.bpm;
            }
        d.plane_w[c] = d.mcus_x * comp_h[c] * 8;
        d.plane_h[c] = d.mcus_y * comp_v[c] * 8;
        d.real_w[c] = (d.width * comp_h[c] + d.hmax - 1) / d.hmax;
        d.real_h[c] = (d.height * comp_v[c] + d.vmax - 1) / d.vmax;
        d.plane_off[c] = int(off);
        off += size_t(d.plane_w[c]) * size_t(d.plane_h[c]);  // <= 3 x 16400^2 < 2^31
    }
    for (int c = d.ncomp; c < 3; ++c) {
        no_codes_slot(d, c);
        no_codes_slot(d, 3 + c);
    }
    d.comp_packed = 0;
    for (int b = 0; b < d.bpm; ++b) d.comp_packed |= uint32_t(d.blk_comp[b]) << (2 * b);
    std::memcpy(d.nat, kNatural, 64);
    plan.n_mcus = size_t(d.mcus_x) * d.mcus_y;
    if (plan.n_mcus * d.bpm > (size_t(1) << 25)) {
        why = "image too large";
        return 2;
    }
    d.total_blocks = int(plan.n_mcus * d.bpm);
    // a block costs the entropy coder at least two bits (a DC code and an end-of-block code of one bit each): a header that
    // promises more blocks than the rest of the file could hold is a decompression bomb or a truncated file - either way
    // nothing is allocated for it (scratch is sized from total_blocks: 128 bytes each)
    if ((n - p) * 4 < size_t(d.total_blocks)) {
        why = "the file is too short for the frame its header declares";
        return 2;
    }
    plan.n_seg = plan.ri ? int((plan.n_mcus + plan.ri - 1) / plan.ri) : 1;
    d.seg_blocks = plan.ri ? plan.ri * d.bpm : d.total_blocks;
    d.fancy = (d.ncomp == 3 && d.hmax == 2 && d.real_w[1] > 2) ? 1 : 0;  // jdsample.c: fancy only when downsampled_width > 2
    plan.scan_begin = p;
    return 0;
}

MvlmJpegPlan* mvlm_jpeg_plan_new() { return new MvlmJpegPlan(); }
void mvlm_jpeg_plan_delete(MvlmJpegPlan* p) { delete p; }
void mvlm_jpeg_plan_dims(const MvlmJpegPlan& plan, int* width, int* height, int* components) {
    if (width) *width = plan.dev.width;
    if (height) *height = plan.dev.height;
    if (components) *components = plan.dev.ncomp;
}

// upper bound of what mvlm_jpeg_fill_stage writes
size_t mvlm_jpeg_stage_bytes(const MvlmJpegPlan& plan, size_t n) {
    // (a restart interval costs up to one subsequence of padding; there cannot be more intervals than two-byte markers)
    const size_t intervals = std::min(size_t(plan.n_seg), (n - plan.scan_begin) / 2 + 1);
    const size_t stream = (n - plan.scan_begin) + intervals * SUB_BYTES + 4 * SUB_BYTES;
    return 256 + (sizeof(JpegDev) + 255) / 256 * 256 + (stream + 255) / 256 * 256 + (stream / SUB_BYTES + 8) * 4;
}

// header + stream without byte stuffing (every restart interval starts on a subsequence boundary, padded with 1-bits) +
// restart interval of every subsequence.  0: ok; 2: the markers in the entropy-coded segment are not what the header says
int mvlm_jpeg_fill_stage(MvlmJpegPlan& plan, const uint8_t* data, size_t n, uint8_t* stage, std::string& why) {
    const size_t off_hdr = 0;
    plan.off_stream = (sizeof(JpegDev) + 255) / 256 * 256;
    uint8_t* out = stage + plan.off_stream;
    size_t o = 0;
    size_t p = plan.scan_begin;
    int seg = 0;
    std::vector<int32_t> seg_first;  // first subsequence of every restart interval
    seg_first.push_back(0);
    for (;;) {
        const uint8_t* q = p < n ? static_cast<const uint8_t*>(std::memchr(data + p, 0xFF, n - p)) : nullptr;
        if (!q || size_t(q - data) + 1 >= n) {
            std::memcpy(out + o, data + p, n - p);  // (a stream that ends without EOI: libjpeg also decodes what is there)
            o += n - p;
            break;
        }
        const size_t run = size_t(q - (data + p));
        std::memcpy(out + o, data + p, run);
        o += run;
        const int nxt = q[1];
        p = size_t(q - data);
        if (nxt == 0) {
            out[o++] = 0xFF;
            p += 2;
        } else if (nxt >= 0xD0 && nxt <= 0xD7) {
            if (++seg >= plan.n_seg) {
                why = "more restart markers than restart intervals";
                return 2;
            }
            const size_t pad = (SUB_BYTES - o % SUB_BYTES) % SUB_BYTES;
            std::memset(out + o, 0xFF, pad);
            o += pad;
            if (int32_t(o / SUB_BYTES) == seg_first.back()) {
                why = "an empty restart interval";
                return 2;
            }
            seg_first.push_back(int32_t(o / SUB_BYTES));
            p += 2;
        } else if (nxt == 0xFF) {
            p += 1;
        } else {
            break;  // EOI (or any other marker): the scan ends here
        }
    }
    if (seg + 1 != plan.n_seg) {
        why = "fewer restart markers than restart intervals";
        return 2;
    }
    if (int32_t((o + SUB_BYTES - 1) / SUB_BYTES) == seg_first.back()) {
        why = seg ? "an empty restart interval" : "empty scan";
        return 2;
    }
    const size_t pad = (SUB_BYTES - o % SUB_BYTES) % SUB_BYTES + 2 * SUB_BYTES;
    std::memset(out + o, 0xFF, pad);
    o += pad;
    plan.stream_bytes = o;
    plan.dev.n_sub = int(o / SUB_BYTES) - 2;  // (the two subsequences of fill are only ever read, never decoded)
    if (plan.dev.n_sub <= 0) {
        why = "empty scan";
        return 2;
    }
    plan.off_subseg = plan.off_stream + (o + 255) / 256 * 256;
    auto* sub_seg = reinterpret_cast<int32_t*>(stage + plan.off_subseg);
    seg_first.push_back(plan.dev.n_sub);
    for (int g = 0; g < plan.n_seg; ++g)
        for (int s = seg_first[size_t(g)]; s < seg_first[size_t(g) + 1]; ++s) sub_seg[s] = g;
    plan.stage_used = plan.off_subseg + size_t(plan.dev.n_sub) * 4;
    std::memcpy(stage + off_hdr, &plan.dev, sizeof(JpegDev));
    return 0;
}

