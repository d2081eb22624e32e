// Baseline JPEG -> RGB texture on the GPU (the texture half of the reference's mesh ingest: vtkJPEGReader in
// src/mvlm/utils/utils3d.py:28-34 / :42-48 / :457-462; the decoder behind it is libjpeg-turbo inside VTK - JDCT_ISLOW,
// fancy upsampling, JCS_RGB).  Results are byte for byte libjpeg's (oracle/jpeg.py restates the arithmetic; the tests pin
// both against Pillow's libjpeg-turbo).
//
// Huffman decoding is serial per restart interval, 16 ms of one core for a 2048 x 2048 texture.  Here the entropy-coded
// segment is cut into 1024-bit subsequences, one thread each.  A thread needs the decoder state at its first bit (bit
// offset of the first whole symbol, block within the MCU, zigzag index); it starts from a guess, and because Huffman codes
// self-synchronise the state it reaches at its END is soon the true one whatever it started from.  Rounds of "decode my
// subsequence from the state my predecessor ended in" run until a whole round changes no state - then every state is the
// true one BY CONSTRUCTION (the first subsequence of every restart interval starts from the known state, the rest follows
// by induction), however many rounds that took; a thread whose start state did not change since the last round keeps its
// stored result.  A segmented prefix sum over the subsequences' block counts and DC-difference sums then tells every
// thread where its coefficients go and what its DC predictors are, one more pass writes the coefficients, and the inverse
// DCT, the chroma upsampling and the colour conversion are plain data-parallel kernels.
//
// Scope: SOF0 / SOF1, 8 bit, one interleaved scan, grey or YCbCr with 1x1 / 2x1 / 2x2 luma over 1x1 chroma, restart
// intervals.  Everything else is reported as "not taken" (return code 2) and the caller decodes with libjpeg on the host.
#include <algorithm>
#include <cstddef>
#include <cstring>

#include "common.h"

namespace {

constexpr int SUB_WORDS = 32;               // 32-bit words per subsequence
constexpr int SUB_BITS = SUB_WORDS * 32;    // 1024
constexpr int SUB_BYTES = SUB_WORDS * 4;    // 128
constexpr int DEC_WG = 256;                 // subsequences (= threads) per workgroup of the decode kernels
constexpr int DEC_PITCH = DEC_WG + 1;       // LDS row pitch of the transposed stream words (odd: no bank pattern)
constexpr int LUT_BITS = 12;                // codes up to this length decode with one table read
constexpr int N_LONG = 16 - LUT_BITS + 1;   // lengths LUT_BITS .. 16: the limits the longer codes are found with
constexpr int N_SLOTS = 6;                  // Huffman tables on the device: DC of component c in slot c, AC in slot 3 + c
constexpr int FIRST_BATCH = 12, ROUNDS_PER_BATCH = 24;
constexpr int MAX_ROUNDS = FIRST_BATCH + 20 * ROUNDS_PER_BATCH;
constexpr uint32_t STATE_DEAD = 0x80000000u;

const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// what the kernels read (built on the host, copied with the stream)
struct JpegDev {
    int width, height, ncomp, hmax, vmax, mcus_x, mcus_y, bpm;
    int n_sub, total_blocks, seg_blocks, fancy;
    int blk_comp[8], blk_bx[8], blk_by[8];
    int plane_w[3], plane_h[3], real_w[3], real_h[3], plane_off[3];
    uint32_t comp_packed;                  // component of block b of the MCU in bits [2b, 2b + 1]
    uint16_t quant[3][64];                 // natural order
    alignas(16) uint16_t lut[N_SLOTS][1 << LUT_BITS];  // length << 8 | symbol for codes of at most LUT_BITS bits, else 0
    // Canonical codes, left-aligned to 16 bits: limit[i] = one past the last code of length LUT_BITS + i (= the first code
    // of the next length).  A 16-bit window w that misses the table has length LUT_BITS + #{i : w >= limit[i]}, and its
    // symbol is vals[valptr[i] + ((w - limit[i - 1]) >> (16 - length))] with i = length - LUT_BITS.
    uint32_t limit[N_SLOTS][8];
    uint8_t valptr[N_SLOTS][8];
    uint8_t vals[N_SLOTS][256];
    uint8_t nat[64];
};
static_assert(N_LONG <= 8, "limit table");

struct HuffHost {
    bool have = false;
    uint8_t counts[16];
    uint8_t vals[256];
    int n = 0;
};

}  // namespace

struct MvlmJpegPlan {
    JpegDev dev;
    int ri = 0, n_seg = 0;
    size_t scan_begin = 0;
    size_t n_mcus = 0;
    // after mvlm_jpeg_fill_stage
    size_t off_stream = 0, stream_bytes = 0, off_subseg = 0, stage_used = 0;
};

namespace {

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
    int off = 0;
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
        d.plane_off[c] = off;
        off += d.plane_w[c] * d.plane_h[c];
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

namespace {

struct DecLds {
    uint32_t words[SUB_WORDS * DEC_PITCH];
    alignas(16) uint16_t lut[N_SLOTS][1 << LUT_BITS];
    uint32_t limit[N_SLOTS][8];
    uint8_t valptr[N_SLOTS][8];
    uint8_t vals[N_SLOTS][256];
    uint8_t nat[64];
};

// 16-byte loads, all of a thread's loads in flight before the first store (a loop of dependent 4-byte load -> store pairs
// on one wave cost 70 us per launch - more than the decoding itself)
__device__ inline void dec_stage(DecLds& L, const JpegDev* hdr, const uint32_t* stream, long total_words, int sub0) {
    const int t = threadIdx.x;
    // this workgroup's DEC_WG subsequences and the one behind them, word j of subsequence c at [j][c]
    const long q0 = long(sub0) * (SUB_WORDS / 4);
    const long total_quads = total_words / 4;
    const uint4* src4 = reinterpret_cast<const uint4*>(stream);
    constexpr int QUADS = SUB_WORDS / 4 * DEC_PITCH;
    constexpr int Q_ITERS = (QUADS + DEC_WG - 1) / DEC_WG;
    uint4 v[Q_ITERS];
#pragma unroll
    for (int it = 0; it < Q_ITERS; ++it) {
        const int i = it * DEC_WG + t;
        const long g = q0 + i;
        v[it] = (i < QUADS && g < total_quads) ? src4[g] : make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    constexpr int TAB_QUADS = (sizeof(L.lut) + sizeof(L.limit) + sizeof(L.valptr) + sizeof(L.vals) + sizeof(L.nat)) / 16;
    static_assert((sizeof(L.lut) + sizeof(L.limit) + sizeof(L.valptr) + sizeof(L.vals) + sizeof(L.nat)) % 16 == 0, "table size");
    static_assert(offsetof(DecLds, nat) - offsetof(DecLds, lut) == offsetof(JpegDev, nat) - offsetof(JpegDev, lut), "table layout");
    static_assert(offsetof(JpegDev, lut) % 16 == 0 && offsetof(DecLds, lut) % 16 == 0 && offsetof(DecLds, limit) % 16 == 0, "table alignment");
    constexpr int T_ITERS = (TAB_QUADS + DEC_WG - 1) / DEC_WG;
    const uint4* tsrc = reinterpret_cast<const uint4*>(hdr->lut);
    uint4* tdst = reinterpret_cast<uint4*>(L.lut);
    uint4 tv[T_ITERS];
#pragma unroll
    for (int it = 0; it < T_ITERS; ++it) {
        const int i = it * DEC_WG + t;
        tv[it] = i < TAB_QUADS ? tsrc[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < Q_ITERS; ++it) {
        const int i = it * DEC_WG + t;
        if (i < QUADS) {
            const int c = i / (SUB_WORDS / 4), j = (i % (SUB_WORDS / 4)) * 4;
            L.words[(j + 0) * DEC_PITCH + c] = __builtin_bswap32(v[it].x);
            L.words[(j + 1) * DEC_PITCH + c] = __builtin_bswap32(v[it].y);
            L.words[(j + 2) * DEC_PITCH + c] = __builtin_bswap32(v[it].z);
            L.words[(j + 3) * DEC_PITCH + c] = __builtin_bswap32(v[it].w);
        }
    }
#pragma unroll
    for (int it = 0; it < T_ITERS; ++it) {
        const int i = it * DEC_WG + t;
        if (i < TAB_QUADS) tdst[i] = tv[it];
    }
    __syncthreads();
}

// One subsequence from the state `in` (bit offset | block in MCU << 8 | zigzag index << 16) to the first symbol that
// starts behind its last bit.  WRITE: the coefficients go to coef (absolute block `ablk`, at most up to `blk_limit`),
// dc0..2 are the running predictors; otherwise they collect the sums of the DC differences, n_blk the blocks completed.
template <bool WRITE>
__device__ inline uint32_t dec_subsequence(const DecLds& L, int t, uint32_t in, int bpm, uint32_t comp_packed, int& n_blk, int& dc0,
                                           int& dc1, int& dc2, int16_t* coef, long ablk, long blk_limit) {
    int pos = int(in & 0xFF), blk = int((in >> 8) & 0xFF), k = int((in >> 16) & 0xFF);
    auto word = [&](int j) { return L.words[(j & (SUB_WORDS - 1)) * DEC_PITCH + t + (j >> 5)]; };
    int w0 = pos >> 5;
    uint32_t hi = word(w0), lo = word(w0 + 1), nx = word(w0 + 2);  // nx: read one word ahead, off the critical path
    n_blk = 0;
    if (WRITE && ablk >= blk_limit) return STATE_DEAD;
    // The loop is one dependent chain per lane (window -> table entry -> lengths -> next window) and the kernel lasts as
    // long as its slowest lane, so the body is written with selects instead of branches wherever both sides are cheap.
    while (pos < SUB_BITS) {
        const int w = pos >> 5;
        const bool adv = w != w0;  // (a symbol is at most 27 bits: the window moves by at most one word)
        hi = adv ? lo : hi;
        lo = adv ? nx : lo;
        nx = word(w + 2);
        w0 = w;
        const int sh = pos & 31;
        const uint32_t peek = sh ? __builtin_amdgcn_alignbit(hi, lo, 32 - sh) : hi;  // the 32 bits at pos
        const int comp = int((comp_packed >> (2 * blk)) & 3u);
        const bool dc_mode = k == 0;
        const int slot = dc_mode ? comp : 3 + comp;
        const uint32_t e = L.lut[slot][peek >> (32 - LUT_BITS)];
        int len = int(e >> 8), sym = int(e & 255);
        if (len == 0) {
            // A code longer than the table's index: its length from the left-aligned limits, its symbol from the offset
            // to the first code of that length.
            const uint4 la = *reinterpret_cast<const uint4*>(&L.limit[slot][0]);
            const uint4 lb = *reinterpret_cast<const uint4*>(&L.limit[slot][4]);
            const uint2 vp = *reinterpret_cast<const uint2*>(&L.valptr[slot][0]);
            const uint32_t lim[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
            const uint32_t w16 = peek >> 16;
            int i = 0;
            uint32_t first = 0;
#pragma unroll
            for (int j = 0; j < N_LONG; ++j)
                if (w16 >= lim[j]) {
                    i = j + 1;
                    first = lim[j];
                }
            if (i >= N_LONG) return STATE_DEAD;  // no code starts with these bits
            len = LUT_BITS + i;
            const uint32_t ptr = ((i < 4 ? vp.x : vp.y) >> (8 * (i & 3))) & 255u;
            sym = L.vals[slot][(ptr + ((w16 - first) >> (16 - len))) & 255u];
        }
        const int s = sym & 15, r = sym >> 4;
        // the s bits behind the code, sign-extended by T.81 F.2.2.1 (s == 0: no bits, value 0)
        const uint32_t bits = uint32_t(uint64_t(peek << len) >> (32 - s));
        int val = int(bits) - (bits < ((1u << s) >> 1) ? (1 << s) - 1 : 0);
        pos += len + s;
        // DC: the difference, position 0.  AC: run r then a coefficient, or (s == 0) sixteen zeros / the end of the block.
        const int kpos = dc_mode ? 0 : k + r;
        const bool eob = !dc_mode && s == 0 && r != 15;
        const bool coefficient = dc_mode || s != 0;
        // (selects, not an indexed array: "the predictor of component comp" becomes scratch memory otherwise - a scratch
        // round trip per block)
        const int dv = dc_mode ? val : 0;
        dc0 += comp == 0 ? dv : 0;
        dc1 += comp == 1 ? dv : 0;
        dc2 += comp == 2 ? dv : 0;
        val = dc_mode ? (comp == 0 ? dc0 : (comp == 1 ? dc1 : dc2)) : val;
        if (WRITE && coefficient && kpos < 64) coef[uint32_t(ablk) * 64u + uint32_t(kpos)] = int16_t(val);  // zigzag order
        k = eob ? 64 : kpos + 1;
        const bool done = k >= 64;
        k = done ? 0 : k;
        blk = done ? (blk + 1 == bpm ? 0 : blk + 1) : blk;
        n_blk += done ? 1 : 0;
        if (WRITE) {
            ablk += done ? 1 : 0;
            if (ablk >= blk_limit) return STATE_DEAD;
        }
    }
    return uint32_t(pos - SUB_BITS) | (uint32_t(blk) << 8) | (uint32_t(k) << 16);
}

__global__ __launch_bounds__(DEC_WG) void jpeg_sync_kernel(const JpegDev* hdr, const uint32_t* stream, long total_words,
                                                          const int32_t* sub_seg, const uint32_t* in_prev, const uint32_t* in_cur,
                                                          uint32_t* in_next, uint32_t* out_state, int32_t* out_nblk, int32_t* out_dc,
                                                          int32_t* changed, int round) {
    if (round > 0 && changed[round - 1] == 0) return;  // the round before changed nothing: every state is final
    __shared__ DecLds L;
    const int sub0 = blockIdx.x * DEC_WG;
    const int n_sub = hdr->n_sub;
    const int t = threadIdx.x, s = sub0 + t;
    // does any thread of this workgroup have to decode?
    const bool mine = s < n_sub;
    const uint32_t in = mine ? in_cur[s] : 0;
    const bool need = mine && (round == 0 || in != in_prev[s]);
    if (__syncthreads_or(need)) {
        dec_stage(L, hdr, stream, total_words, sub0);
        if (need) {
            int n_blk, dc0 = 0, dc1 = 0, dc2 = 0;
            const uint32_t o = dec_subsequence<false>(L, t, in, hdr->bpm, hdr->comp_packed, n_blk, dc0, dc1, dc2, nullptr, 0, 0);
            out_state[s] = o;
            out_nblk[s] = n_blk;
            out_dc[3 * s] = dc0;
            out_dc[3 * s + 1] = dc1;
            out_dc[3 * s + 2] = dc2;
        }
    }
    if (!mine) return;
    const int g = sub_seg[s];
    if (s == 0 || sub_seg[s - 1] != g) in_next[s] = 0;  // the first subsequence of a restart interval: the known state
    if (s + 1 < n_sub && sub_seg[s + 1] == g) {
        const uint32_t o = out_state[s];
        const uint32_t cur = in_cur[s + 1];
        const uint32_t nxt = (o & STATE_DEAD) ? cur : o;
        in_next[s + 1] = nxt;
        if (nxt != cur) atomicOr(&changed[round], 1);
    }
}

// Segmented exclusive prefix sums over the subsequences of every restart interval - blocks completed and DC-difference
// sums before each subsequence - in two levels: jpeg_scan_kernel scans SCAN_WG subsequences per workgroup (exclusive
// values that are final wherever a restart interval began inside the workgroup, "open" = 1 where the workgroup's carry-in
// still has to be added), jpeg_scan_carry_kernel scans the workgroups' totals.
constexpr int SCAN_WG = 256;
constexpr int SCAN_MAX_GROUPS = 1024;

// (f, v) <- (f_left, v_left) (+) (f, v) = (f_left | f, f ? v : v_left + v): inclusive Hillis-Steele scan in LDS
template <int N>
__device__ inline void segmented_scan(int (&f)[N], int (&v)[N][4], int t) {
    for (int d = 1; d < N; d <<= 1) {
        int pf = 0, pv[4] = {0, 0, 0, 0};
        const bool has = t >= d;
        if (has) {
            pf = f[t - d];
            for (int i = 0; i < 4; ++i) pv[i] = v[t - d][i];
        }
        __syncthreads();
        if (has) {
            if (!f[t])
                for (int i = 0; i < 4; ++i) v[t][i] += pv[i];
            f[t] |= pf;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_WG) void jpeg_scan_kernel(const JpegDev* hdr, const int32_t* sub_seg, const int32_t* out_nblk,
                                                           const int32_t* out_dc, int32_t* excl, int32_t* open, int32_t* group_total,
                                                           int32_t* group_flag) {
    __shared__ int f[SCAN_WG];
    __shared__ int v[SCAN_WG][4];
    const int n_sub = hdr->n_sub, t = threadIdx.x, s = blockIdx.x * SCAN_WG + t;
    const bool live = s < n_sub;
    const int first = live && (s == 0 || sub_seg[s] != sub_seg[s - 1]) ? 1 : 0;
    const int own[4] = {live ? out_nblk[s] : 0, live ? out_dc[3 * s] : 0, live ? out_dc[3 * s + 1] : 0, live ? out_dc[3 * s + 2] : 0};
    f[t] = first;
    for (int i = 0; i < 4; ++i) v[t][i] = own[i];
    __syncthreads();
    segmented_scan<SCAN_WG>(f, v, t);
    if (live) {
        for (int i = 0; i < 4; ++i) excl[4 * s + i] = first ? 0 : v[t][i] - own[i];
        open[s] = f[t] ? 0 : 1;
    }
    if (t == SCAN_WG - 1) {
        for (int i = 0; i < 4; ++i) group_total[4 * blockIdx.x + i] = v[t][i];
        group_flag[blockIdx.x] = f[t];
    }
}

__global__ __launch_bounds__(SCAN_MAX_GROUPS) void jpeg_scan_carry_kernel(const int32_t* group_total, const int32_t* group_flag,
                                                                         int32_t* carry, int n_groups) {
    __shared__ int f[SCAN_MAX_GROUPS];
    __shared__ int v[SCAN_MAX_GROUPS][4];
    const int t = threadIdx.x;
    f[t] = t < n_groups ? group_flag[t] : 0;
    for (int i = 0; i < 4; ++i) v[t][i] = t < n_groups ? group_total[4 * t + i] : 0;
    __syncthreads();
    segmented_scan<SCAN_MAX_GROUPS>(f, v, t);
    if (t < n_groups)
        for (int i = 0; i < 4; ++i) carry[4 * t + i] = t ? v[t - 1][i] : 0;  // what enters workgroup t (inclusive up to t - 1)
}

// short_flag: a restart interval that holds fewer blocks than it must (a truncated or damaged stream)
__global__ __launch_bounds__(DEC_WG) void jpeg_write_kernel(const JpegDev* hdr, const uint32_t* stream, long total_words,
                                                           const int32_t* sub_seg, const uint32_t* in_state, const int32_t* excl,
                                                           const int32_t* open, const int32_t* carry, const int32_t* out_nblk,
                                                           int16_t* coef, int32_t* short_flag) {
    __shared__ DecLds L;
    const int sub0 = blockIdx.x * DEC_WG;
    dec_stage(L, hdr, stream, total_words, sub0);
    const int t = threadIdx.x, s = sub0 + t, n_sub = hdr->n_sub;
    if (s >= n_sub) return;
    const int g = sub_seg[s];
    const long seg0 = long(g) * hdr->seg_blocks;
    const long limit = min(seg0 + hdr->seg_blocks, long(hdr->total_blocks));
    int base[4];
    const int add = open[s];
    for (int i = 0; i < 4; ++i) base[i] = excl[4 * s + i] + (add ? carry[4 * (s / SCAN_WG) + i] : 0);
    if (s + 1 == n_sub || sub_seg[s + 1] != g) {
        if (base[0] + out_nblk[s] < limit - seg0) atomicOr(short_flag, 1);
    }
    int n_blk, dc0 = base[1], dc1 = base[2], dc2 = base[3];
    (void)dec_subsequence<true>(L, t, in_state[s], hdr->bpm, hdr->comp_packed, n_blk, dc0, dc1, dc2, coef, seg0 + base[0], limit);
}

// jidctint.c jpeg_idct_islow in 32-bit integers: one pass over eight values, DESCALE by SHIFT
template <int SHIFT>
__device__ inline void idct8(const int x[8], int y[8]) {
    int z1 = (x[2] + x[6]) * 4433;
    const int tmp2 = z1 - x[6] * 15137;
    const int tmp3 = z1 + x[2] * 6270;
    const int tmp0 = (x[0] + x[4]) * 8192;
    const int tmp1 = (x[0] - x[4]) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = x[7], t1 = x[5], t2 = x[3], t3 = x[1];
    z1 = t0 + t3;
    int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int z5 = (z3 + z4) * 9633;
    t0 *= 2446;
    t1 *= 16819;
    t2 *= 25172;
    t3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 = z3 * -16069 + z5;
    z4 = z4 * -3196 + z5;
    t0 += z1 + z3;
    t1 += z2 + z4;
    t2 += z2 + z3;
    t3 += z1 + z4;
    constexpr int H = 1 << (SHIFT - 1);
    y[0] = (tmp10 + t3 + H) >> SHIFT;
    y[7] = (tmp10 - t3 + H) >> SHIFT;
    y[1] = (tmp11 + t2 + H) >> SHIFT;
    y[6] = (tmp11 - t2 + H) >> SHIFT;
    y[2] = (tmp12 + t1 + H) >> SHIFT;
    y[5] = (tmp12 - t1 + H) >> SHIFT;
    y[3] = (tmp13 + t0 + H) >> SHIFT;
    y[4] = (tmp13 - t0 + H) >> SHIFT;
}

constexpr int IDCT_BLOCKS = 32;  // blocks per workgroup (8 threads each)
constexpr int WS_PITCH = 72;

__global__ __launch_bounds__(IDCT_BLOCKS * 8) void jpeg_idct_kernel(const JpegDev* hdr, const int16_t* coef, uint8_t* planes) {
    __shared__ int16_t cs[IDCT_BLOCKS * 64];
    __shared__ int ws[IDCT_BLOCKS * WS_PITCH];
    __shared__ uint16_t qs[3 * 64];
    const int t = threadIdx.x;
    const long b0 = long(blockIdx.x) * IDCT_BLOCKS;
    const int total = hdr->total_blocks;
    {
        // the decoder wrote zigzag order: the natural position on the way into LDS
        const uint32_t* src = reinterpret_cast<const uint32_t*>(coef + b0 * 64);
        const long avail = (long(total) - b0) * 32;  // words
        for (int i = t; i < IDCT_BLOCKS * 32; i += IDCT_BLOCKS * 8) {
            const uint32_t v = i < avail ? src[i] : 0u;
            const int bl = i >> 5, z = (i & 31) * 2;
            cs[bl * 64 + hdr->nat[z]] = int16_t(v & 0xFFFFu);
            cs[bl * 64 + hdr->nat[z + 1]] = int16_t(v >> 16);
        }
        if (t < 3 * 64) qs[t] = (&hdr->quant[0][0])[t];
    }
    __syncthreads();
    const int bl = t >> 3, i = t & 7;
    const long b = b0 + bl;
    const bool live = b < total;
    const int bi = live ? int(b % hdr->bpm) : 0;
    const long mcu = live ? b / hdr->bpm : 0;
    const int comp = hdr->blk_comp[bi];
    {
        int x[8], y[8];
        for (int r = 0; r < 8; ++r) x[r] = int(cs[bl * 64 + r * 8 + i]) * int(qs[comp * 64 + r * 8 + i]);
        idct8<13 - 2>(x, y);
        for (int r = 0; r < 8; ++r) ws[bl * WS_PITCH + r * 8 + i] = y[r];
    }
    __syncthreads();
    if (!live) return;
    int x[8], y[8];
    for (int c = 0; c < 8; ++c) x[c] = ws[bl * WS_PITCH + i * 8 + c];
    idct8<13 + 2 + 3>(x, y);
    uint32_t lo = 0, hi = 0;
    for (int c = 0; c < 4; ++c) {
        lo |= uint32_t(min(255, max(0, y[c] + 128))) << (8 * c);
        hi |= uint32_t(min(255, max(0, y[c + 4] + 128))) << (8 * c);
    }
    const int my = int(mcu / hdr->mcus_x), mx = int(mcu % hdr->mcus_x);
    const int hc = comp == 0 ? hdr->hmax : 1, vc = comp == 0 ? hdr->vmax : 1;
    const int px = (mx * hc + hdr->blk_bx[bi]) * 8, py = (my * vc + hdr->blk_by[bi]) * 8 + i;
    uint8_t* dst = planes + hdr->plane_off[comp] + long(py) * hdr->plane_w[comp] + px;
    *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
}

// chroma upsampling (jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample, or replication for the narrow images libjpeg
// does not filter) + jdcolor.c ycc_rgb_convert; four pixels per thread
__global__ __launch_bounds__(256) void jpeg_colour_kernel(const JpegDev* hdr, const uint8_t* planes, uint8_t* rgb) {
    const int W = hdr->width, H = hdr->height;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x0 >= W || y >= H) return;
    const uint8_t* Y = planes + hdr->plane_off[0] + long(y) * hdr->plane_w[0];
    int r[4], g[4], b[4];
    if (hdr->ncomp == 1) {
        for (int i = 0; i < 4; ++i) r[i] = g[i] = b[i] = Y[min(x0 + i, hdr->plane_w[0] - 1)];
    } else {
        const int pw = hdr->plane_w[1], dw = hdr->real_w[1], dh = hdr->real_h[1];
        const uint8_t* CB = planes + hdr->plane_off[1];
        const uint8_t* CR = planes + hdr->plane_off[2];
        int cb[4], cr[4];
        if (hdr->hmax == 1) {
            for (int i = 0; i < 4; ++i) {
                const int x = min(x0 + i, dw - 1);
                cb[i] = CB[long(y) * pw + x];
                cr[i] = CR[long(y) * pw + x];
            }
        } else if (!hdr->fancy) {
            const int cy = hdr->vmax == 2 ? y >> 1 : y;
            for (int i = 0; i < 4; ++i) {
                const int x = min((x0 + i) >> 1, dw - 1);
                cb[i] = CB[long(cy) * pw + x];
                cr[i] = CR[long(cy) * pw + x];
            }
        } else if (hdr->vmax == 1) {
            const int c0 = x0 >> 1;
            for (int pl = 0; pl < 2; ++pl) {
                const uint8_t* P = (pl ? CR : CB) + long(y) * pw;
                int* o = pl ? cr : cb;
                const int a = P[max(c0 - 1, 0)], m0 = P[min(c0, dw - 1)], m1 = P[min(c0 + 1, dw - 1)], z = P[min(c0 + 2, dw - 1)];
                o[0] = (3 * m0 + a + 1) >> 2;
                o[1] = (3 * m0 + m1 + 2) >> 2;
                o[2] = (3 * m1 + m0 + 1) >> 2;
                o[3] = (3 * m1 + z + 2) >> 2;
            }
        } else {
            const int c0 = x0 >> 1, cy = y >> 1;
            const int oy = (y & 1) ? min(cy + 1, dh - 1) : max(cy - 1, 0);
            for (int pl = 0; pl < 2; ++pl) {
                const uint8_t* P0 = (pl ? CR : CB) + long(cy) * pw;
                const uint8_t* P1 = (pl ? CR : CB) + long(oy) * pw;
                int* o = pl ? cr : cb;
                const int xa = max(c0 - 1, 0), xm0 = min(c0, dw - 1), xm1 = min(c0 + 1, dw - 1), xz = min(c0 + 2, dw - 1);
                const int a = 3 * P0[xa] + P1[xa], m0 = 3 * P0[xm0] + P1[xm0], m1 = 3 * P0[xm1] + P1[xm1], z = 3 * P0[xz] + P1[xz];
                o[0] = (3 * m0 + a + 8) >> 4;
                o[1] = (3 * m0 + m1 + 7) >> 4;
                o[2] = (3 * m1 + m0 + 8) >> 4;
                o[3] = (3 * m1 + z + 7) >> 4;
            }
        }
        for (int i = 0; i < 4; ++i) {
            const int yy = Y[min(x0 + i, hdr->plane_w[0] - 1)];
            const int u = cb[i] - 128, v = cr[i] - 128;
            r[i] = min(255, max(0, yy + ((91881 * v + 32768) >> 16)));
            g[i] = min(255, max(0, yy + ((-22554 * u + 32768 - 46802 * v) >> 16)));
            b[i] = min(255, max(0, yy + ((116130 * u + 32768) >> 16)));
        }
    }
    uint8_t* out = rgb + (long(y) * W + x0) * 3;
    if ((W & 3) == 0) {
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
        o32[0] = uint32_t(r[0]) | uint32_t(g[0]) << 8 | uint32_t(b[0]) << 16 | uint32_t(r[1]) << 24;
        o32[1] = uint32_t(g[1]) | uint32_t(b[1]) << 8 | uint32_t(r[2]) << 16 | uint32_t(g[2]) << 24;
        o32[2] = uint32_t(b[2]) | uint32_t(r[3]) << 8 | uint32_t(g[3]) << 16 | uint32_t(b[3]) << 24;
    } else {
        for (int i = 0; i < 4 && x0 + i < W; ++i) {
            out[3 * i] = uint8_t(r[i]);
            out[3 * i + 1] = uint8_t(g[i]);
            out[3 * i + 2] = uint8_t(b[i]);
        }
    }
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// Device work of one decode on `stream`: the staged blob goes up, the kernels run, rgb_dev [H, W, 3] is complete when the
// stream has passed.  The host waits for the stream twice (the convergence check of the synchronisation rounds, the "every
// restart interval is complete" check).  0: done; 1: HIP failure (ctx->fail); 2: the stream did not decode (why).
// Caller holds ctx->upload_mu (the scratch below belongs to the upload path).
int mvlm_jpeg_run(mvlm_ctx* ctx, MvlmJpegPlan& plan, const uint8_t* stage_pinned, uint8_t* rgb_dev, hipStream_t stream,
                  std::string& why, int* rounds_out) {
    const JpegDev& d = plan.dev;
    const size_t n_sub = size_t(d.n_sub);
    size_t plane_bytes = 0;
    for (int c = 0; c < d.ncomp; ++c) plane_bytes += size_t(d.plane_w[c]) * d.plane_h[c];
    // scratch layout
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += up256(bytes);
        return o;
    };
    const size_t o_blob = take(plan.stage_used);
    const size_t o_in = take(3 * n_sub * 4);
    const size_t o_out = take(n_sub * 4), o_nblk = take(n_sub * 4), o_dc = take(3 * n_sub * 4), o_base = take(4 * n_sub * 4);
    const size_t n_groups = (n_sub + SCAN_WG - 1) / SCAN_WG;
    if (n_groups > size_t(SCAN_MAX_GROUPS)) {
        why = "entropy-coded segment too long";
        return 2;
    }
    const size_t o_open = take(n_sub * 4), o_group = take(n_groups * (4 + 1 + 4) * 4);
    const size_t o_flags = take((MAX_ROUNDS + 8) * 4);
    const size_t o_coef = take(size_t(d.total_blocks) * 128);
    const size_t o_planes = take(plane_bytes + 16);
    if (ctx->jpeg_scratch_cap < off) {
        if (ctx->jpeg_scratch) {
            MVLM_CHECK_HIP(ctx, hipStreamSynchronize(stream));
            (void)hipFree(ctx->jpeg_scratch);
            ctx->jpeg_scratch = nullptr;
            ctx->jpeg_scratch_cap = 0;
        }
        const size_t cap = off + off / 4;
        MVLM_CHECK_HIP(ctx, hipMalloc(&ctx->jpeg_scratch, cap));
        ctx->jpeg_scratch_cap = cap;
    }
    if (!ctx->jpeg_flags_host) MVLM_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->jpeg_flags_host), (MAX_ROUNDS + 8) * 4, hipHostMallocDefault));
    auto* base = static_cast<uint8_t*>(ctx->jpeg_scratch);
    const auto* hdr = reinterpret_cast<const JpegDev*>(base + o_blob);
    const auto* words = reinterpret_cast<const uint32_t*>(base + o_blob + plan.off_stream);
    const long total_words = long(plan.stream_bytes / 4);
    const auto* sub_seg = reinterpret_cast<const int32_t*>(base + o_blob + plan.off_subseg);
    auto* in3 = reinterpret_cast<uint32_t*>(base + o_in);
    auto* out_state = reinterpret_cast<uint32_t*>(base + o_out);
    auto* out_nblk = reinterpret_cast<int32_t*>(base + o_nblk);
    auto* out_dc = reinterpret_cast<int32_t*>(base + o_dc);
    auto* prefix = reinterpret_cast<int32_t*>(base + o_base);
    auto* open = reinterpret_cast<int32_t*>(base + o_open);
    auto* group_total = reinterpret_cast<int32_t*>(base + o_group);
    auto* group_flag = group_total + 4 * n_groups;
    auto* carry = group_flag + n_groups;
    auto* flags = reinterpret_cast<int32_t*>(base + o_flags);  // [round] "changed", [MAX_ROUNDS] "short"
    auto* coef = reinterpret_cast<int16_t*>(base + o_coef);
    uint8_t* planes = base + o_planes;

    MVLM_CHECK_HIP(ctx, hipMemcpyAsync(base + o_blob, stage_pinned, plan.stage_used, hipMemcpyHostToDevice, stream));
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(in3, 0, 3 * n_sub * 4, stream));
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(flags, 0, (MAX_ROUNDS + 8) * 4, stream));
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(coef, 0, size_t(d.total_blocks) * 128, stream));
    const unsigned dec_grid = unsigned((n_sub + DEC_WG - 1) / DEC_WG);
    int final_round = -1;
    // a batch of rounds, then the host looks at the "changed" flags; rounds behind the final one return at once (5 us
    // each), so the first batch is short (most textures need 6-12 rounds) and the later ones longer
    for (int r0 = 0, batch = FIRST_BATCH; r0 < MAX_ROUNDS && final_round < 0; r0 += batch, batch = ROUNDS_PER_BATCH) {
        for (int r = r0; r < r0 + batch; ++r) {
            const uint32_t* prev = in3 + size_t((r + 2) % 3) * n_sub;
            const uint32_t* cur = in3 + size_t(r % 3) * n_sub;
            uint32_t* next = in3 + size_t((r + 1) % 3) * n_sub;
            hipLaunchKernelGGL(jpeg_sync_kernel, dim3(dec_grid), dim3(DEC_WG), 0, stream, hdr, words, total_words, sub_seg, prev, cur,
                               next, out_state, out_nblk, out_dc, flags, r);
        }
        MVLM_CHECK_HIP(ctx, hipGetLastError());
        MVLM_CHECK_HIP(ctx, hipMemcpyAsync(ctx->jpeg_flags_host + r0, flags + r0, size_t(batch) * 4, hipMemcpyDeviceToHost, stream));
        MVLM_CHECK_HIP(ctx, hipStreamSynchronize(stream));
        for (int r = r0; r < r0 + batch; ++r)
            if (ctx->jpeg_flags_host[r] == 0) {
                final_round = r;
                break;
            }
    }
    if (rounds_out) *rounds_out = final_round;
    if (final_round < 0) {
        why = "the subsequence decoders did not synchronise";
        return 2;
    }
    const uint32_t* in_final = in3 + size_t(final_round % 3) * n_sub;
    hipLaunchKernelGGL(jpeg_scan_kernel, dim3(unsigned(n_groups)), dim3(SCAN_WG), 0, stream, hdr, sub_seg, out_nblk, out_dc, prefix, open,
                       group_total, group_flag);
    hipLaunchKernelGGL(jpeg_scan_carry_kernel, dim3(1), dim3(SCAN_MAX_GROUPS), 0, stream, group_total, group_flag, carry, int(n_groups));
    hipLaunchKernelGGL(jpeg_write_kernel, dim3(dec_grid), dim3(DEC_WG), 0, stream, hdr, words, total_words, sub_seg, in_final, prefix, open,
                       carry, out_nblk, coef, flags + MAX_ROUNDS);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(unsigned((d.total_blocks + IDCT_BLOCKS - 1) / IDCT_BLOCKS)), dim3(IDCT_BLOCKS * 8), 0, stream,
                       hdr, coef, planes);
    hipLaunchKernelGGL(jpeg_colour_kernel, dim3(unsigned((d.width + 1023) / 1024), unsigned(d.height)), dim3(256), 0, stream, hdr, planes, rgb_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    MVLM_CHECK_HIP(ctx, hipMemcpyAsync(ctx->jpeg_flags_host + MAX_ROUNDS, flags + MAX_ROUNDS, 4, hipMemcpyDeviceToHost, stream));
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(stream));
    if (ctx->jpeg_flags_host[MAX_ROUNDS]) {
        why = "the entropy-coded segment ends before the image does";
        return 2;
    }
    return 0;
}
