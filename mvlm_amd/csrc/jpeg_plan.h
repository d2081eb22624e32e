// Host side of the device JPEG decoder (jpeg.hip): the layout the kernels read (JpegDev), the plan of one decode, and the
// parser / stager that fill them from the file's bytes.  Plain C++ - no HIP - so that jpeg_plan.cpp also builds with
// g++ -fsanitize=address,undefined for tests/test_jpeg_plan_sanitized.py: this is the code that reads UNTRUSTED bytes.
#ifndef MVLM_JPEG_PLAN_H
#define MVLM_JPEG_PLAN_H

#include <cstddef>
#include <cstdint>
#include <string>

namespace mvlm_jpeg {

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

}  // namespace mvlm_jpeg

struct MvlmJpegPlan {
    using JpegDev = mvlm_jpeg::JpegDev;
    JpegDev dev;
    int ri = 0, n_seg = 0;
    size_t scan_begin = 0;
    size_t n_mcus = 0;
    // after mvlm_jpeg_fill_stage
    size_t off_stream = 0, stream_bytes = 0, off_subseg = 0, stage_used = 0;
};


// 0: a JPEG the device decoder takes; 2: not taken (why says which rule); never touches the GPU
int mvlm_jpeg_plan_impl(const uint8_t* data, size_t n, MvlmJpegPlan& plan, std::string& why);
void mvlm_jpeg_plan_dims(const MvlmJpegPlan& plan, int* width, int* height, int* components);
MvlmJpegPlan* mvlm_jpeg_plan_new();
void mvlm_jpeg_plan_delete(MvlmJpegPlan* p);
// upper bound of what mvlm_jpeg_fill_stage writes
size_t mvlm_jpeg_stage_bytes(const MvlmJpegPlan& plan, size_t n);
int mvlm_jpeg_fill_stage(MvlmJpegPlan& plan, const uint8_t* data, size_t n, uint8_t* stage, std::string& why);

#endif
