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
#include "jpeg_plan.h"

using namespace mvlm_jpeg;

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
static int jpeg_run_impl(mvlm_ctx* ctx, MvlmJpegPlan& plan, const uint8_t* stage_pinned, uint8_t* rgb_dev, hipStream_t stream,
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

int mvlm_jpeg_run(mvlm_ctx* ctx, MvlmJpegPlan& plan, const uint8_t* stage_pinned, uint8_t* rgb_dev, hipStream_t stream,
                  std::string& why, int* rounds_out) {
    const int rc = jpeg_run_impl(ctx, plan, stage_pinned, rgb_dev, stream, why, rounds_out);
    // 0 and 2 return with the stream drained (see above).  A HIP failure may leave copies and kernels queued that still read the
    // staging slot and write rgb_dev, and the callers hand both back to their pools next: wait here, whatever the wait returns.
    if (rc == 1) (void)hipStreamSynchronize(stream);
    // The scratch is grow-only for the textures a scanner writes (~50 MB at 3546 x 2282); one oversized file (the parser admits up
    // to 16384 x 16384 = 2.6 GB of scratch) does not keep its allocation for the life of the context.
    if (ctx->jpeg_scratch && ctx->jpeg_scratch_cap > (size_t(512) << 20)) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(ctx->jpeg_scratch);
        ctx->jpeg_scratch = nullptr;
        ctx->jpeg_scratch_cap = 0;
    }
    return rc;
}
