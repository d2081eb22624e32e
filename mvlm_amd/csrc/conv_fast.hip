// OPT-IN reduced-cost arithmetic for the big 3x3 layers ("fast" precision; never the default, never the headline).
//
// The exact path multiplies fp32 by fp32 on v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD).  Here every fp32 operand is
// split into three bf16 terms, x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 significant
// bits together), and the product is evaluated as the six cross terms of weight >= 2^-16
//     xh*wh + xh*wm + xm*wh + xh*wl + xm*wm + xl*wh          (dropped: xm*wl, xl*wm, xl*wl <= 2^-23 relative)
// on v_mfma_f32_32x32x16_bf16 (1024 FLOP/clk/SIMD) with fp32 accumulation: 6/16 of the exact path's matrix time for
// a result whose error is of the order of the fp32 rounding of the sum itself - but NOT bit-identical to it, so an
// argmax near-tie can flip (bench.py reports the measured flip rate; the reference's own precedent is
// torch.set_float32_matmul_precision("medium"), paulsenpredictor.py:175-176, which is much coarser).
//
// Kernel: implicit GEMM like conv_mfma_kernel (32-pixel row tiles, same accumulator layout, so the same kind of
// epilogue), one 512-thread workgroup per CU; tiles 128 channels x 8 rows or 64 channels x 16 rows:
//   * 8 waves = 2 (output-channel halves) x 4 (groups of TRI / 4 pixel rows); a wave owns MT x NT = 4 MFMA tiles
//   * K runs in units of (16 input channels) x (one row of 3 taps).  LDS per stage:
//       X  [k-half 2][split 3][(TRI + 2) x 34 haloed pixels][8 channels] bf16   32 640 / 58 752 B  (restaged once per 3 units)
//       W  [tap 3][k-half 2][split 3][COUT_T][8 channels] bf16                  288 * COUT_T B
//     both double-buffered (139 / 154 KB): while unit u multiplies, the global loads of unit u + 1 are in flight and
//     their BatchNorm / split / LDS writes are issued between the MFMAs of unit u; one barrier per unit.
//     A lane's MFMA fragment (8 consecutive channels of one pixel / one output channel) is one ds_read_b128, and
//     consecutive lanes read consecutive 16-byte slots: conflict-free.
//   * the consumer-side BatchNorm + ReLU is applied in fp32 before the split, zero padding after it.
//   * weights are split and laid out on the host ([chunk][tap row][tap][k-half][split][cout_pad][8], mvlm_pack_fast_weights)
//     so that staging W is a linear 16-byte copy.
// Second form, NS = 2 ("fast16"): every operand as TWO fp16 terms, x = h + l (h = fp16(x), l = fp16(x - h): 22 significant
// bits), the product as the three cross terms xh*wh + xh*wl + xl*wh (dropped: xl*wl <= 2^-22 relative) on
// v_mfma_f32_32x32x16_f16 - half the matrix work of the bf16x3 form (3/16 of the exact path's).  fp16 products of 11-bit
// mantissas are exact in fp32, so the error per product is the operands' 2^-22 representation error plus the dropped
// term: 3-4 x that of bf16x3, still below what the order of an fp32 summation over K = 2304 terms moves.  fp16 has a
// narrow exponent: the weights are scaled by a power of two per layer on the host (largest |w| into [2^13, 2^14); the
// epilogue multiplies by the inverse, exactly), activations go in as they are: |x| >= 65504 has no fp16 form - the kernel
// raises the context's flag word, the executor turns that pass's maxima into NaN and the caller repeats the pass on bf16x3
// (predictor / pipeline).  Small activations lose nothing that matters: |x - (h + l)| <= max(2^-22 |x|, 2^-25).
// Layers it serves: 3x3, input channels a multiple of 16, output channels a multiple of 64, width a multiple of 32,
// height of 8, plain NCHW output with optional bias / post-BN+ReLU / raw copy / one residual.  Everything else (and
// every layer in the default "exact" precision) runs on conv_mfma_kernel.
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

namespace {

constexpr int FT_TW = 32, FT_PW = FT_TW + 2;
constexpr int FT_THREADS = 512;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Two values at a time (one 32-bit register per split term; element 0 in the low half):
// x -> (h, m, l) bf16 with h + m + l == x up to 2^-24 |x| (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned* h, unsigned* m, unsigned* l) {
    *h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(*h << 16), r1 = x1 - __uint_as_float(*h & 0xFFFF0000u);
    *m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(*m << 16), s1 = r1 - __uint_as_float(*m & 0xFFFF0000u);
    *l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
}
// x -> (h, l) fp16 with |x - (h + l)| <= max(2^-22 |x|, 2^-25); |x| >= 65520 gives h = +-inf (the epilogue sees what that does)
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned* h, unsigned* l) {
    const f16x2 hh = __builtin_convertvector((f32x2){x0, x1}, f16x2);  // round to nearest even
    // x - float(h), exactly that (one rounding), as an fma so that the fp16 operand converts inside the instruction
    const float r0 = __builtin_fmaf(static_cast<float>(hh.x), -1.f, x0), r1 = __builtin_fmaf(static_cast<float>(hh.y), -1.f, x1);
    *h = __builtin_bit_cast(unsigned, hh);
    *l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, f16x2));
}

// the arithmetic of a split: NS terms per operand, the cross terms (weight term, activation term) smallest first
template <int NS>
struct Split;
template <>
struct Split<3> {
    using frag = bf16x8;
    static constexpr int NTERMS = 6;
    static constexpr int WS[6] = {0, 2, 1, 0, 1, 0}, XS[6] = {2, 0, 1, 1, 0, 0};  // (h,l) (l,h) (m,m) | (h,m) (m,h) | (h,h)
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Split<2> {
    using frag = f16x8;
    static constexpr int NTERMS = 3;
    static constexpr int WS[3] = {0, 1, 0}, XS[3] = {1, 0, 0};  // (h,l) (l,h) | (h,h)
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// COUT_T output channels x (TRI rows x 32 pixels); 8 waves = WM channel groups x WN row groups (2 x 4; the 32-channel tile
// of the f16x2 form: 1 x 8 - conv2.conv2 / conv2.conv3 / conv3.conv2 / conv3.conv3 of the stem, 4.2 ms per 96 views on exact tiles)
template <int COUT_T, int TRI, int NS = 3>
struct FastCfg {
    static constexpr int WM = COUT_T >= 64 ? 2 : 1, WN = 8 / WM;
    static constexpr int MT = COUT_T / (32 * WM);        // 32-row MFMA tiles per wave (channels)
    static constexpr int NT = TRI / WN;                   // 32-pixel row segments per wave
    static constexpr int PH = TRI + 2, NPIX = FT_PW * PH;  // haloed tile
    static constexpr int X_BYTES = 2 * NS * NPIX * 16;      // [k-half][split][pixel][8 ch]
    static constexpr int W_BYTES = 3 * 2 * NS * COUT_T * 16;  // one tap row: [tap][k-half][split][cout][8 ch]
    static constexpr int W_ITEMS = W_BYTES / 16, W_ITERS = (W_ITEMS + FT_THREADS - 1) / FT_THREADS;
    static constexpr int X_ITEMS = 2 * NPIX, X_ITERS = (X_ITEMS + FT_THREADS - 1) / FT_THREADS;
    static constexpr int TAP_MFMAS = Split<NS>::NTERMS * MT * NT;  // per tap and wave
    static constexpr size_t LDS = size_t(2) * X_BYTES + size_t(2) * W_BYTES + 4 * 256 * sizeof(float);  // + BatchNorm table + as many zeros
    static_assert(LDS <= 160 * 1024, "stages must fit the CU's LDS");
    static_assert(TAP_MFMAS == (NS == 3 ? 24 : 12) || (NS == 2 && TAP_MFMAS == 6),
                  "the staging schedules below are written for 24 (bf16x3) / 12 (f16x2) / 6 (f16x2, 32 channels) MFMAs per tap");
    static_assert(X_ITERS <= 3 && W_ITERS <= 5, "staging schedule");
    static_assert(NT % 2 == 0, "the pooled output pairs the row segments of a wave");
};

// GEN: the rarely needed parts (input channels that do not fill the last chunk, a second residual) are compiled in
template <int COUT_T, int TRI, bool GEN, int NS>
__global__ __launch_bounds__(FT_THREADS, 2) void conv_split_kernel(const ConvArgs a, const unsigned short* __restrict__ wq,
                                                                  const int tiles_x, const int tiles_y, const int cout_tiles) {
    using C = FastCfg<COUT_T, TRI, NS>;
    using SP = Split<NS>;
    using frag_t = typename SP::frag;
    constexpr int TAPM = C::TAP_MFMAS;
    constexpr int MT = C::MT, NT = C::NT, NPIX = C::NPIX, X_BYTES = C::X_BYTES, W_BYTES = C::W_BYTES;
    constexpr int W_ITEMS = C::W_ITEMS, W_ITERS = C::W_ITERS, X_ITEMS = C::X_ITEMS, X_ITERS = C::X_ITERS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fast[];
    unsigned char* const sX = smem_fast;                        // 2 stages
    unsigned char* const sW = smem_fast + 2 * X_BYTES;          // 2 stages
    float* const sbn = reinterpret_cast<float*>(sW + 2 * W_BYTES);  // [128 channel pairs][scale, scale, shift, shift], then 512 zeros

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave % C::WM, wn = wave / C::WM;

    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ct = lid % cout_tiles, pt = lid / cout_tiles;
    const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, b0 = pt / (tiles_x * tiles_y);
    const int x0 = tx * FT_TW, y0 = ty * TRI, co0 = ct * COUT_T;
    const int H = a.H, W = a.W;
    const unsigned HW = unsigned(H) * unsigned(W);
    const int n_chunks = a.cin_pad / 16;
    const bool partial_cin = GEN && a.cin != a.cin_pad;
    const bool has_bn = a.pre_scale != nullptr;

    // ---- staging plans ------------------------------------------------------------------------------------------
    // X: item e -> (k-half kh, haloed pixel p); eight channels 16 chunk + 8 kh + j of that pixel
    // Every thread stages X_ITERS items whether or not that many exist: an item past the end repeats the thread's item 0
    // (same loads, same LDS writes - no conditional code in the main loop), a pixel outside the image reads its BatchNorm
    // parameters from the table of zeros (0 * x + 0, whatever was loaded from the clamped address: zero padding).
    unsigned xoff[X_ITERS];   // element offset of channel (8 kh) of chunk 0 at that pixel
    int xdst[X_ITERS];        // byte offset inside an X stage of split 0 (+ NPIX * 16 per split)
    int xbn[X_ITERS];         // float offset of the item's BatchNorm parameters for chunk 0 in sbn
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
        const int e0 = tid + i * FT_THREADS, e = e0 < X_ITEMS ? e0 : tid;
        const int kh = e / NPIX, p = e - kh * NPIX;
        const int yy = p / FT_PW, xx = p - yy * FT_PW;
        const int y = y0 + yy - 1, x = x0 + xx - 1;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        xoff[i] = unsigned(b0 * a.in_ctot + a.in_coff + 8 * kh) * HW + (ok ? unsigned(y * W + x) : 0u);
        xdst[i] = (kh * NS * NPIX + p) * 16;
        xbn[i] = 16 * kh + (ok ? 0 : 512);
    }
    // W: item f -> segment (tap, kh, split) and output channel c of the tile; linear in the host layout
    unsigned wsrc[W_ITERS];   // u16 element offset inside a (chunk, tap row) block of the packed weights
    int wdst[W_ITERS];
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
        const int f0 = tid + i * FT_THREADS, f = f0 < W_ITEMS ? f0 : tid % W_ITEMS;  // (the 32-channel tile has fewer items than threads)
        const int seg = f / COUT_T, c = f - seg * COUT_T;
        wsrc[i] = unsigned((seg * a.cout_pad + co0 + c) * 8);
        wdst[i] = f * 16;
    }
    const size_t w_block = size_t(6 * NS) * a.cout_pad * 8;  // u16 elements of one (chunk, tap row) block

    float xv[X_ITERS][8];      // activations in flight / waiting for their split
    u32x4 xq[NS];              // one item's eight channels as 16-bit pairs, per split
    u32x4 wv[W_ITERS];         // weights in flight
    const float relu_floor = has_bn ? 0.f : -__builtin_inff();  // no BatchNorm in front: no ReLU either
    const int last_unit = 3 * n_chunks - 1;

    // ---- staging micro-operations (each one small enough for the shadow of one or two MFMAs).  None of them is conditional
    // on "is there a next stage": past the end they load the last stage again and write the LDS stage nobody reads any more.
    auto load_w = [&](int unit) __attribute__((always_inline)) {
        const unsigned short* const wb = wq + size_t(unit < last_unit ? unit : last_unit) * w_block;
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i)
#if defined(MVLM_FAST_ABLATE_NO_W)  // timing experiment only
            if (unit < 2)
#endif
            wv[i] = *reinterpret_cast<const u32x4*>(wb + wsrc[i]);
    };
    auto store_w = [&](auto ic, int unit) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
#if defined(MVLM_FAST_ABLATE_NO_W)
        if (unit < 2)
#endif
        *reinterpret_cast<u32x4*>(sW + (unit & 1) * W_BYTES + wdst[i]) = wv[i];
    };
    auto load_x_item = [&](auto ic, int chunk_) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < X_ITERS) {
            const int chunk = chunk_ < n_chunks ? chunk_ : n_chunks - 1;
            const float* const base = a.in + size_t(chunk) * 16 * HW;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = !partial_cin || chunk * 16 + ((xbn[i] & 16) >> 1) + j < a.cin;
#if defined(MVLM_FAST_ABLATE_NO_XLOAD)  // timing experiment only
                if (chunk_ < 2)
#endif
                xv[i][j] = base[ok ? xoff[i] + unsigned(j) * HW : 0u];  // what is not there is zeroed in convert_pair
            }
        }
    };
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        static_for<0, X_ITERS>([&](auto ic) { load_x_item(ic, chunk); });
    };
    // BatchNorm + ReLU, split: channels 2 jp and 2 jp + 1 of item i -> one u32 per split
    auto convert_pair = [&](auto ic, auto jc, int chunk) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, jp = decltype(jc)::value;
        if constexpr (i < X_ITERS) {
            const int c = chunk * 16 + ((xbn[i] & 16) >> 1) + 2 * jp;  // (xbn = 16 kh [+ 512])
            const f32x4 bn = *reinterpret_cast<const f32x4*>(sbn + chunk * 32 + xbn[i] + 4 * jp);  // scale, scale, shift, shift of channels c, c + 1
            float v0 = fmaxf(fmaf(xv[i][2 * jp], bn.x, bn.z), relu_floor);
            float v1 = fmaxf(fmaf(xv[i][2 * jp + 1], bn.y, bn.w), relu_floor);
            if (partial_cin) {  // channels past cin
                v0 = c < a.cin ? v0 : 0.f;
                v1 = c + 1 < a.cin ? v1 : 0.f;
            }
#if defined(MVLM_FAST_ABLATE_NO_CONVERT)  // timing experiment only
            xq[0][jp] = __float_as_uint(xv[i][2 * jp]);
            xq[1][jp] = __float_as_uint(xv[i][2 * jp + 1]);
            if constexpr (NS == 3) xq[NS - 1][jp] = __float_as_uint(xv[i][2 * jp]) ^ 1u;
            return;
#endif
            if constexpr (NS == 3) {
                unsigned h, m, l;
                split3_pair(v0, v1, &h, &m, &l);
                xq[0][jp] = h;
                xq[1][jp] = m;
                xq[2][jp] = l;
            } else {
                unsigned h, l;
                split2_pair(v0, v1, &h, &l);
                xq[0][jp] = h;
                xq[1][jp] = l;
            }
        }
    };
    auto store_x = [&](auto ic, auto sc, int chunk) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, sp = decltype(sc)::value;
        if constexpr (i < X_ITERS) {
#if defined(MVLM_FAST_ABLATE_NO_XSTORE)  // timing experiment only
            if (xdst[i] == -2)
#endif
            *reinterpret_cast<u32x4*>(sX + (chunk & 1) * X_BYTES + xdst[i] + sp * NPIX * 16) = xq[sp];
        }
    };
    // all of item i at once (prologue)
    auto stage_item = [&](auto ic, int chunk) __attribute__((always_inline)) {
        static_for<0, 4>([&](auto jc) { convert_pair(ic, jc, chunk); });
        static_for<0, NS>([&](auto sc) { store_x(ic, sc, chunk); });
    };

    // ---- accumulators and per-lane LDS offsets ----------------------------------------------------------------------
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // B fragment of pixel row n of this wave, tap (dy, dx): haloed pixel ((NT wn + n + dy) * PW + l31 + dx), k-half = half
    int boff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) boff[n] = (half * NS * NPIX + (NT * wn + n) * FT_PW + l31) * 16;
    // A fragment of MFMA tile m: output channel wm * 32 MT + 32 m + l31, k-half = half
    int aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = (half * NS * COUT_T + wm * 32 * MT + 32 * m + l31) * 16;  // (wm < WM)

    // ---- prologue: BatchNorm table, chunk 0's activations, unit 0's weights ----------------------------------------------
    // [channel pair][scale, scale, shift, shift]; without BatchNorm: 1, 0 (and no ReLU: relu_floor); then the zeros
    for (int i = tid; i < 256; i += FT_THREADS) {  // padding channels: relu(0 * 0 + 0) = 0
        const bool there = has_bn && i < a.cin;
        sbn[4 * (i >> 1) + (i & 1)] = there ? a.pre_scale[i] : (has_bn || i >= a.cin_pad ? 0.f : 1.f);
        sbn[4 * (i >> 1) + 2 + (i & 1)] = there ? a.pre_shift[i] : 0.f;
        sbn[512 + i] = 0.f;
        sbn[768 + i] = 0.f;
    }
    load_w(0);
    load_x(0);
    __syncthreads();  // BatchNorm table visible
    static_for<0, W_ITERS>([&](auto ic) { store_w(ic, 0); });
    static_for<0, X_ITERS>([&](auto ic) { stage_item(ic, 0); });
    // (in the order the main loop leaves its loads in flight at the end of a chunk: the wait counters the compiler places
    // at the top of the loop then fit both ways in)
    load_x_item(std::integral_constant<int, 0>{}, 1);
    load_w(1);
    static_for<1, X_ITERS>([&](auto ic) { load_x_item(ic, 1); });
    __syncthreads();

    // One unit = (chunk, tap row ROW): 3 taps x 24 (12) MFMAs per wave.  The staging of what comes next is cut into
    // micro-operations placed BETWEEN the MFMAs (slot q = 24 tap + index): the matrix pipe holds a wave's vector issue
    // for 8 of an MFMA's 32 cycles, the rest is where BatchNorm / split / LDS writes of the next stage run.
    // Global loads run TWO stages ahead, each into the registers the previous stage has just been converted / stored from,
    // so a load has a unit or more (the activations: three) to arrive before anything waits for it:
    //   every unit : LDS writes of the next unit's weights at q = 24 + 4 i (tap 1), then the loads of the unit after it
    //   ROW == 1   : item 0 of the next chunk: pairs at q = 2, 8, 14, 20, then its loads of the chunk after; splits stored
    //                at q = 44, 50, 56
    //   ROW == 2   : item 1: pairs at q = 2, 6, 10, 14 (+ loads), stores at q = 18, 20, 22; item 2: pairs at q = 28, 32, 36, 40
    //                (+ loads), stores at q = 60, 64, 68                                     (72-slot table; the 36-slot one below)
    auto unit = [&](auto rowc, int chunk) __attribute__((always_inline)) {
        constexpr int ROW = decltype(rowc)::value;
        const int u = 3 * chunk + ROW;
        const unsigned char* const xs = sX + (chunk & 1) * X_BYTES;
        const unsigned char* const ws = sW + (u & 1) * W_BYTES;
        frag_t af[MT][NS], bf[NT][NS];
        static_for<0, 3>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int poff = (ROW * FT_PW + t) * 16;  // tap (ROW, t): dy = ROW, dx = t in the haloed tile
#if defined(MVLM_FAST_ABLATE_ONE_TAP)  // timing experiment only: fragments are read for the first tap of a unit only
            if constexpr (t == 0)
#endif
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m][sp] = *reinterpret_cast<const frag_t*>(ws + ((t * 2 * NS + sp) * COUT_T) * 16 + aoff[m]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bf[n][sp] = *reinterpret_cast<const frag_t*>(xs + sp * NPIX * 16 + boff[n] + poff);
            }
            // the cross terms, smallest first (Split<NS>)
            static_for<0, TAPM>([&](auto ic) {
                constexpr int i = decltype(ic)::value, q = TAPM * t + i;
                constexpr int p = i / (MT * NT), mn = i % (MT * NT), m = mn / NT, n = mn % NT;
                acc[m][n] = SP::mfma(af[m][SP::WS[p]], bf[n][SP::XS[p]], acc[m][n]);
                // ---- side work of slot q ----
#if defined(MVLM_FAST_ABLATE_NO_STAGING)  // timing experiment only: wrong results
                if constexpr (q < 0) {
#else
                if constexpr (q >= 0) {
#endif
                // slot tables: bf16x3 has 72 slots per unit, f16x2 has 36 - the same micro-operations (one split fewer) at
                // half the distance.  W stores in tap 1; item 0 during ROW 1; items 1 and 2 during ROW 2; an item's splits are
                // stored after its four channel pairs have been converted, the next item's conversion starts after that.
                // (TAPM == 6, the 32-channel tile: 18 slots, one weight item, three activation items)
                constexpr bool T6 = TAPM == 6;
                constexpr int WQ0 = T6 ? 6 : (NS == 3 ? 24 : 12), WQS = T6 ? 1 : (NS == 3 ? 4 : 2);
                if constexpr (q >= WQ0 && q < WQ0 + WQS * W_ITERS && (q - WQ0) % WQS == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    store_w(std::integral_constant<int, (q - WQ0) / WQS>{}, u + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // Loads complete in order: a wait for the weights also waits for every OLDER load.  The activations come from HBM
                // and take several times as long as the weights (L2), so in every unit the weight loads are issued first and
                // the activation loads behind them - a weight wait then leaves them in flight.
                constexpr int WL = WQ0 + WQS * (W_ITERS - 1) + 1, XL = WL + 1;
                // (the scheduler is told to leave the loads and the weights' LDS writes in their slots: left alone it sinks
                // the loads to the end of the unit, which halves the time they have to arrive)
                if constexpr (q == WL) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_w(u + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int C0 = T6 ? 1 : (NS == 3 ? 2 : 1), C0S = T6 ? 1 : (NS == 3 ? 6 : 3);      // ROW 1, item 0: pairs at C0 + C0S j
                constexpr int S0 = T6 ? 10 : (NS == 3 ? 44 : 20), S0S = T6 ? 2 : (NS == 3 ? 6 : 4);    //               stores at S0 + S0S sp
                constexpr int C1 = T6 ? 1 : (NS == 3 ? 2 : 1), C1S = T6 ? 1 : (NS == 3 ? 4 : 2);      // ROW 2, item 1
                constexpr int S1 = T6 ? 5 : (NS == 3 ? 18 : 9), S1S = T6 ? 1 : (NS == 3 ? 2 : 1);
                constexpr int C2 = T6 ? 9 : (NS == 3 ? 28 : 18), C2S = T6 ? 1 : (NS == 3 ? 4 : 2);    // ROW 2, item 2
                constexpr int S2 = T6 ? 14 : (NS == 3 ? 60 : 28), S2S = T6 ? 1 : (NS == 3 ? 4 : 2);
                if constexpr (ROW == 1) {
                    if constexpr (q >= C0 && q < C0 + 4 * C0S && (q - C0) % C0S == 0) {
                        convert_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, (q - C0) / C0S>{}, chunk + 1);
                    }
                    if constexpr (q == (C0 + 3 * C0S + 1 > XL ? C0 + 3 * C0S + 1 : XL)) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_x_item(std::integral_constant<int, 0>{}, chunk + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (q >= S0 && q < S0 + NS * S0S && (q - S0) % S0S == 0) {
                        store_x(std::integral_constant<int, 0>{}, std::integral_constant<int, (q - S0) / S0S>{}, chunk + 1);
                    }
                }
                if constexpr (ROW == 2 && X_ITERS >= 2) {
                    if constexpr (q >= C1 && q < C1 + 4 * C1S && (q - C1) % C1S == 0) {
                        convert_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, (q - C1) / C1S>{}, chunk + 1);
                    }
                    if constexpr (q == (C1 + 3 * C1S + 1 > XL ? C1 + 3 * C1S + 1 : XL)) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_x_item(std::integral_constant<int, 1>{}, chunk + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (q >= S1 && q < S1 + NS * S1S && (q - S1) % S1S == 0) {
                        store_x(std::integral_constant<int, 1>{}, std::integral_constant<int, (q - S1) / S1S>{}, chunk + 1);
                    }
                }
                if constexpr (ROW == 2 && X_ITERS >= 3) {
                    if constexpr (q >= C2 && q < C2 + 4 * C2S && (q - C2) % C2S == 0) {
                        convert_pair(std::integral_constant<int, 2>{}, std::integral_constant<int, (q - C2) / C2S>{}, chunk + 1);
                    }
                    if constexpr (q == (C2 + 3 * C2S + 1 > XL + 1 ? C2 + 3 * C2S + 1 : XL + 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_x_item(std::integral_constant<int, 2>{}, chunk + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (q >= S2 && q < S2 + NS * S2S && (q - S2) % S2S == 0) {
                        store_x(std::integral_constant<int, 2>{}, std::integral_constant<int, (q - S2) / S2S>{}, chunk + 1);
                    }
                }
                }
            });
        });
#if !defined(MVLM_FAST_ABLATE_NO_BARRIER)  // timing experiment only
        __syncthreads();
#endif
    };
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        unit(std::integral_constant<int, 0>{}, chunk);
        unit(std::integral_constant<int, 1>{}, chunk);
        unit(std::integral_constant<int, 2>{}, chunk);
    }

    // ---- epilogue: bias, post-BatchNorm + ReLU, raw copy, residual, store -----------------------------------------------
    // One MFMA tile row (32 channels x NT row segments) at a time: all its residual values are requested first, then
    // the stores follow; the layer kinds that carry the time run without per-element feature tests.
    // f16x2: an activation beyond fp16's range went in as +-inf; every product with it, hence every output whose window
    // holds it, is inf or NaN.  That does NOT reliably survive the network (the next layer's ReLU is a v_max, which drops a
    // NaN), so the kernel raises the context's flag word; the executor turns the pass's maxima into NaN when it is set
    // (cnn_graph.hip) and the caller repeats the pass on bf16x3.
    float finite_chk = 0.f;
    // SCAT: the hourglass up-path (paulsenpredictor.py:334-359) - the value goes to its 2x2 block of the skip tensor,
    // added in place; the blocks of four channels are requested together before those channels are finished
    auto epilogue = [&](auto raw_c, auto res_c, auto par_c, auto scat_c) __attribute__((always_inline)) {
        constexpr bool RAW = decltype(raw_c)::value, RES = decltype(res_c)::value, SCAT = decltype(scat_c)::value;
        constexpr int PAR = decltype(par_c)::value;  // 0 none, 1 bias, 2 bias + post-BatchNorm + ReLU, 3 decide at run time
        static_for<0, MT>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const int cob = co0 + wm * 32 * MT + 32 * m + 4 * half;  // + (r & 3) + 8 (r >> 2)
            unsigned pix[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) pix[n] = unsigned((y0 + NT * wn + n) * W + x0 + l31);
            float resv[16][NT];
            if constexpr (RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cob + (r & 3) + 8 * (r >> 2);
                    const float* const p = a.res1 + (size_t(b0) * a.res1_ctot + a.res1_coff + (co < a.cout ? co : 0)) * HW;
#pragma unroll
                    for (int n = 0; n < NT; ++n) resv[r][n] = p[pix[n]];
                }
            }
            // the second residual (conv7: r3 + ll1 + conv7(...)) the same way - one load per output element inside the store
            // loop is a chain of memory latencies
            constexpr bool RES2 = GEN && PAR == 3;
            float resv2[RES2 ? 16 : 1][NT];
            const bool res2 = RES2 && a.res2 != nullptr;
            if constexpr (RES2) {
                if (res2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = cob + (r & 3) + 8 * (r >> 2);
                        const float* const p = a.res2 + (size_t(b0) * a.res2_ctot + a.res2_coff + (co < a.cout ? co : 0)) * HW;
#pragma unroll
                        for (int n = 0; n < NT; ++n) resv2[r][n] = p[pix[n]];
                    }
                }
            }
            unsigned pix2[SCAT ? NT : 1];  // first element of the pixel's 2x2 block in the full-resolution plane
            if constexpr (SCAT) {
#pragma unroll
                for (int n = 0; n < NT; ++n) pix2[n] = unsigned(2 * (y0 + NT * wn + n)) * unsigned(2 * W) + unsigned(2 * (x0 + l31));
            }
            const unsigned W2 = 2u * unsigned(W);
            constexpr int SKB = NT <= 2 ? 4 : 2;  // channels whose skip blocks are in flight together (64 x 16 tile: registers)
            static_for<0, 16 / SKB>([&](auto rbc) {
                constexpr int rb = decltype(rbc)::value;  // accumulator registers SKB rb ... (register r = channel cob + (r & 3) + 8 (r >> 2))
                f32x2 sk0[SCAT ? SKB : 1][NT], sk1[SCAT ? SKB : 1][NT];
                if constexpr (SCAT) {
#pragma unroll
                    for (int q = 0; q < SKB; ++q) {
                        const int r = SKB * rb + q;
                        const int co = cob + (r & 3) + 8 * (r >> 2);
                        const float* const ps = a.skip + (size_t(b0) * a.skip_ctot + a.skip_coff + (co < a.cout ? co : 0)) * (4 * size_t(HW));
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            sk0[q][n] = *reinterpret_cast<const f32x2*>(ps + pix2[n]);
                            sk1[q][n] = *reinterpret_cast<const f32x2*>(ps + pix2[n] + W2);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < SKB; ++q) {
                    const int r = SKB * rb + q;
                    const int co = cob + (r & 3) + 8 * (r >> 2);
                    if (co >= a.cout) continue;
                    float bias = 0.f, ps = 1.f, pt = 0.f;
                    if (PAR == 1 || PAR == 2 || (PAR == 3 && a.bias)) bias = a.bias[co];
                    const bool post = PAR == 2 || (PAR == 3 && a.post_scale);
                    if (post) {
                        ps = a.post_scale[co];
                        pt = a.post_shift[co];
                    }
                    float vf[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if constexpr (NS == 2) finite_chk = fmaf(acc[m][n][r], 0.f, finite_chk);  // stays 0 unless inf / NaN
                        float v = (NS == 2 ? acc[m][n][r] * a.fast_unscale : acc[m][n][r]) + bias;  // f16x2: weights carry a power-of-two scale
                        if (post) v = fmaxf(fmaf(v, ps, pt), 0.f);
                        if (RAW) a.out_raw[(size_t(b0) * a.raw_ctot + a.raw_coff + co) * HW + pix[n]] = v;
                        if constexpr (RES) v += resv[r][n];
                        if constexpr (RES2) {
                            if (res2) v += resv2[r][n];
                        }
                        vf[n] = v;
                        if constexpr (SCAT) {
                            float* const po = a.out + (size_t(b0) * a.out_ctot + a.out_coff + co) * (4 * size_t(HW)) + pix2[n];
                            *reinterpret_cast<f32x2*>(po) = (f32x2){v + sk0[q][n].x, v + sk0[q][n].y};
                            *reinterpret_cast<f32x2*>(po + W2) = (f32x2){v + sk1[q][n].x, v + sk1[q][n].y};
                        } else {
                            if (a.out) a.out[(size_t(b0) * a.out_ctot + a.out_coff + co) * HW + pix[n]] = v;
                        }
                    }
                    if constexpr (!SCAT) {
                        // F.max_pool2d(y, 2, 2) from the epilogue (round 5; the exact tiles always had it): the rows of a pair are
                        // two row segments of this lane (n, n + 1), the columns two neighbouring lanes
                        if (a.pool_out) {  // (wave-uniform)
                            float* const pp = a.pool_out + (size_t(b0) * a.pool_ctot + a.pool_coff + co) * (HW / 4);
#pragma unroll
                            for (int n = 0; n < NT; n += 2) {
                                const float v2 = fmaxf(vf[n], vf[n + 1]);
                                const int vi = __float_as_int(v2);
                                const float other = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xf, 0xf, false));  // lane ^ 1
                                if ((l31 & 1) == 0) pp[unsigned((y0 + NT * wn + n) >> 1) * unsigned(W >> 1) + unsigned((x0 + l31) >> 1)] = fmaxf(v2, other);
                            }
                        }
                    }
                }
            });
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    const bool has_raw = a.out_raw != nullptr, has_res = a.res1 != nullptr;
    const bool plain = !a.bias && !a.post_scale && !a.res2;
    const bool scat = a.up_out == 1;
    if (scat && has_raw)
        epilogue(T_{}, T_{}, std::integral_constant<int, 0>{}, T_{});   // conv1 / conv2 of a level's last block on the way up
    else if (scat)
        epilogue(F_{}, T_{}, std::integral_constant<int, 0>{}, T_{});   // ... its conv3
    else if (has_raw && has_res && plain)
        epilogue(T_{}, T_{}, std::integral_constant<int, 0>{}, F_{});   // residual block conv1 / conv2
    else if (!has_raw && has_res && plain)
        epilogue(F_{}, T_{}, std::integral_constant<int, 0>{}, F_{});   // residual block conv3
    else if (!has_raw && !has_res && a.bias && a.post_scale && !a.res2)
        epilogue(F_{}, F_{}, std::integral_constant<int, 2>{}, F_{});   // conv5 / conv9
    else if (has_raw && has_res)
        epilogue(T_{}, T_{}, std::integral_constant<int, 3>{}, F_{});
    else if (has_raw)
        epilogue(T_{}, F_{}, std::integral_constant<int, 3>{}, F_{});
    else if (has_res)
        epilogue(F_{}, T_{}, std::integral_constant<int, 3>{}, F_{});
    else
        epilogue(F_{}, F_{}, std::integral_constant<int, 3>{}, F_{});
    if constexpr (NS == 2) {
        if (!(finite_chk == 0.f) && a.fast_ovf) *a.fast_ovf = 1u;  // (every writer stores the same word)
    }
}

template <int COUT_T, int TRI, bool GEN, int NS>
int launch_fast(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq) {
    using C = FastCfg<COUT_T, TRI, NS>;
    const int bit = COUT_T == 32 ? 54 + (GEN ? 1 : 0) : (NS == 3 ? 60 : 56) + (COUT_T == 128 ? 0 : 1) + (GEN ? 2 : 0);
    if (!((ctx->conv_attr_mask >> bit) & 1ull)) {
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<COUT_T, TRI, GEN, NS>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS)));
        ctx->conv_attr_mask |= 1ull << bit;
    }
    const int tiles_x = a.W / FT_TW, tiles_y = a.H / TRI, cout_tiles = a.cout_pad / COUT_T;
    const long nblk = long(tiles_x) * tiles_y * a.B * cout_tiles;
    MVLM_REQUIRE(ctx, nblk > 0 && nblk < (1l << 31), "conv_fast: bad grid");
    hipLaunchKernelGGL((conv_split_kernel<COUT_T, TRI, GEN, NS>), dim3((unsigned)nblk), dim3(FT_THREADS), C::LDS, ctx->cur_stream(), a, wq,
                       tiles_x, tiles_y, cout_tiles);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

template <int NS>
int launch_fast_ns(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq_dev) {
    if constexpr (NS == 2) {
        if (a.cout_pad == 32)
            return (a.cin != a.cin_pad || a.res2) ? launch_fast<32, 16, true, 2>(ctx, a, wq_dev) : launch_fast<32, 16, false, 2>(ctx, a, wq_dev);
    }
    if (a.cin != a.cin_pad || a.res2)
        return a.cout_pad % 128 == 0 ? launch_fast<128, 8, true, NS>(ctx, a, wq_dev) : launch_fast<64, 16, true, NS>(ctx, a, wq_dev);
    return a.cout_pad % 128 == 0 ? launch_fast<128, 8, false, NS>(ctx, a, wq_dev) : launch_fast<64, 16, false, NS>(ctx, a, wq_dev);
}

}  // namespace

// can the fast kernel serve this launch?  (its own paddings: mvlm_fast_cin_pad / mvlm_fast_cout_pad, common.h)
bool mvlm_conv_fast_ok(const ConvArgs& a, int splits) {
    if (a.ksize != 3 || !mvlm_fast_channels_ok(a.cin, a.cout)) return false;
    // 128-channel tiles cover 8 rows, 64-channel tiles 16 rows (same matrix work per staged input tile)
    if (mvlm_fast_cout_pad(a.cout) == 32 && splits != 2) return false;  // the 32-channel tile exists in the f16x2 form only
    const int rows = mvlm_fast_cout_pad(a.cout) % 128 == 0 ? 8 : 16;
    // plain NCHW output, or the up-path form: 2x2 scatter into the skip tensor (a residual block's layer: one residual, no
    // bias).  The f16x2 form only: its K loop is short enough for the unoverlapped scatter to pay (rb20.conv1 2089 -> 785 us,
    // the step 59.2 -> 55.0 ms); on bf16x3 the same twelve launches bought 3 % and their different rounding flipped one of
    // 8 064 near-tied argmax planes of the bench's random-weight network, so that form keeps round 2's layer set
    const bool scatter = splits == 2 && a.up_out == 1 && a.skip && a.out && a.res1 && !a.res2 && !a.bias && !a.post_scale;
    // (pooled output, round 5: with the plain NCHW forms; every tile holds an even number of rows per wave)
    return a.W % FT_TW == 0 && a.H % rows == 0 && !a.up_in && ((!a.up_out && !a.skip) || scatter) && !a.amax_val && (!a.pool_out || !a.up_out) &&
           (a.out || a.out_raw || a.pool_out);
}

// splits: 3 = bf16x3 ("fast"), 2 = f16x2 ("fast16": the weights carry the power-of-two scale 1 / unscale)
int mvlm_launch_conv_fast(mvlm_ctx* ctx, const ConvArgs& exact, const unsigned short* wq_dev, int splits, float unscale) {
    MVLM_REQUIRE(ctx, (splits == 3 || splits == 2) && mvlm_conv_fast_ok(exact, splits) && wq_dev, "conv_fast: launch not eligible");
    ConvArgs a = exact;  // the split weights carry their own paddings
    a.fast_ovf = splits == 2 ? ctx->cnn.fast16_flag : nullptr;
    a.cin_pad = mvlm_fast_cin_pad(a.cin);
    a.cout_pad = mvlm_fast_cout_pad(a.cout);
    a.fast_unscale = unscale;
    const double px = double(a.B) * a.H * a.W, lim = 4294967295.0;
    MVLM_REQUIRE(ctx, px * a.in_ctot < lim, "conv_fast: input tensor exceeds 32-bit element offsets (lower the batch)");
    return splits == 3 ? launch_fast_ns<3>(ctx, a, wq_dev) : launch_fast_ns<2>(ctx, a, wq_dev);
}

// ---- host: split + lay out the weights of one 3x3 convolution -----------------------------------------------------------
// w f32 [cout][cin][3][3] -> u16 [cin_pad/16][tap row 3][tap 3][k-half 2][split 3][cout_pad][8]; returns the element count
static inline unsigned short host_bf16(float x) {  // round to nearest even; NaN stays NaN
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float host_bf16_value(unsigned short b) {
    const unsigned u = unsigned(b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

extern "C" size_t mvlm_pack_fast_weights(const float* w, int cout, int cin, int cout_pad, int cin_pad, uint16_t* out) {
    if (!w || cout <= 0 || cin <= 0 || cout_pad < cout || cin_pad < cin || cin_pad % 16 != 0) return 0;
    const size_t total = size_t(cin_pad / 16) * 9 * 2 * 3 * cout_pad * 8;
    if (!out) return total;
    memset(out, 0, total * sizeof(uint16_t));
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const float x = w[(size_t(co) * cin + ci) * 9 + tap];
                const unsigned short h = host_bf16(x);
                const float r1 = x - host_bf16_value(h);
                const unsigned short m = host_bf16(r1);
                const float r2 = r1 - host_bf16_value(m);
                const unsigned short l = host_bf16(r2);
                const int chunk = ci / 16, kh = (ci % 16) / 8, j = ci % 8, row = tap / 3, t = tap % 3;
                const unsigned short parts[3] = {h, m, l};
                for (int s = 0; s < 3; ++s) {
                    const size_t idx = ((((size_t(chunk) * 3 + row) * 3 + t) * 2 + kh) * 3 + s) * cout_pad * 8 + size_t(co) * 8 + j;
                    out[idx] = parts[s];
                }
            }
    return total;
}

// The same for the f16x2 form: w * 2^s as two fp16 terms, s chosen so that the layer's largest |w| lies in [2^13, 2^14)
// (the low term of a weight is then a normal fp16 number down to |w| = 2^-9 of the largest); *unscale = 2^-s.
// u16 [cin_pad/16][tap row 3][tap 3][k-half 2][split 2][cout_pad][8]; returns the element count.
static inline unsigned short host_f16(float x) {
    const _Float16 h = static_cast<_Float16>(x);  // round to nearest even
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}
static inline float host_f16_value(unsigned short b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return static_cast<float>(h);
}

extern "C" size_t mvlm_pack_fast_weights16(const float* w, int cout, int cin, int cout_pad, int cin_pad, uint16_t* out, float* unscale) {
    if (!w || cout <= 0 || cin <= 0 || cout_pad < cout || cin_pad < cin || cin_pad % 16 != 0) return 0;
    const size_t total = size_t(cin_pad / 16) * 9 * 2 * 2 * cout_pad * 8;
    if (!out) return total;
    float amax = 0.f;
    for (size_t i = 0; i < size_t(cout) * cin * 9; ++i) {
        const float v = fabsf(w[i]);
        if (!(v <= 3.0e38f)) return 0;  // NaN / inf weights
        amax = v > amax ? v : amax;
    }
    int s = 0;
    if (amax > 0.f) {
        int e = 0;
        (void)frexpf(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
        s = 14 - e;              // amax * 2^s in [2^13, 2^14)
        s = s > 100 ? 100 : (s < -100 ? -100 : s);
    }
    if (unscale) *unscale = ldexpf(1.f, -s);
    memset(out, 0, total * sizeof(uint16_t));
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const float x = ldexpf(w[(size_t(co) * cin + ci) * 9 + tap], s);
                const unsigned short h = host_f16(x);
                const unsigned short l = host_f16(x - host_f16_value(h));
                const int chunk = ci / 16, kh = (ci % 16) / 8, j = ci % 8, row = tap / 3, t = tap % 3;
                const unsigned short parts[2] = {h, l};
                for (int sp = 0; sp < 2; ++sp) {
                    const size_t idx = ((((size_t(chunk) * 3 + row) * 3 + t) * 2 + kh) * 2 + sp) * cout_pad * 8 + size_t(co) * 8 + j;
                    out[idx] = parts[sp];
                }
            }
    return total;
}

// ---- single convolution through the fast kernel (test hook, mirrors mvlm_conv2d) ---------------------------------------
static int conv2d_fast_impl(mvlm_ctx* ctx, int splits, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                            const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                            const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev) {
    MVLM_REQUIRE(ctx, x_dev && w_host && y_dev, "conv2d_fast: null pointer");
    MVLM_REQUIRE(ctx, mvlm_fast_channels_ok(cin, cout), "conv2d_fast: channel counts outside what the fast kernel serves");
    const int cin_pad = mvlm_fast_cin_pad(cin), cout_pad = mvlm_fast_cout_pad(cout);
    float unscale = 1.f;
    auto pack = [&](uint16_t* out) {
        return splits == 3 ? mvlm_pack_fast_weights(w_host, cout, cin, cout_pad, cin_pad, out)
                           : mvlm_pack_fast_weights16(w_host, cout, cin, cout_pad, cin_pad, out, &unscale);
    };
    std::vector<uint16_t> wq(pack(nullptr));
    MVLM_REQUIRE(ctx, !wq.empty() && pack(wq.data()) == wq.size(), "conv2d_fast: weight packing failed");
    std::vector<float> vec(size_t(2) * cin_pad + size_t(3) * cout_pad, 0.f);
    auto fill = [&](size_t off, const float* src, int n) {
        if (src) memcpy(vec.data() + off, src, size_t(n) * sizeof(float));
    };
    fill(0, pre_scale_host, cin);
    fill(cin_pad, pre_shift_host, cin);
    fill(2 * size_t(cin_pad), bias_host, cout);
    fill(2 * size_t(cin_pad) + cout_pad, post_scale_host, cout);
    fill(2 * size_t(cin_pad) + 2 * size_t(cout_pad), post_shift_host, cout);
    auto* dq = static_cast<uint16_t*>(ctx->get_scratch("conv2d_fast.w", wq.size() * 2));
    auto* dv = static_cast<float*>(ctx->get_scratch("conv2d_fast.v", vec.size() * 4));
    MVLM_REQUIRE(ctx, dq && dv, "conv2d_fast: scratch allocation failed");
    MVLM_CHECK_HIP(ctx, hipMemcpy(dq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
    MVLM_CHECK_HIP(ctx, hipMemcpy(dv, vec.data(), vec.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = x_dev;
    a.in_ctot = cin;
    a.cin = cin;
    a.cin_pad = cin_pad;
    a.B = batch;
    a.H = h;
    a.W = w;
    a.cout = cout;
    a.cout_pad = cout_pad;
    a.ksize = 3;
    a.pre_scale = pre_scale_host ? dv : nullptr;
    a.pre_shift = pre_scale_host ? dv + cin_pad : nullptr;
    a.bias = bias_host ? dv + 2 * size_t(cin_pad) : nullptr;
    a.post_scale = post_scale_host ? dv + 2 * size_t(cin_pad) + cout_pad : nullptr;
    a.post_shift = post_scale_host ? dv + 2 * size_t(cin_pad) + 2 * size_t(cout_pad) : nullptr;
    a.res1 = r_dev;
    a.res1_ctot = cout;
    a.out = y_dev;
    a.out_ctot = cout;
    if (mvlm_launch_conv_fast(ctx, a, dq, splits, unscale)) return 1;
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int mvlm_conv2d_fast(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                                const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                                const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev) {
    MVLM_ENTER(ctx);
    return conv2d_fast_impl(ctx, 3, x_dev, batch, cin, h, w, w_host, cout, bias_host, pre_scale_host, pre_shift_host, post_scale_host,
                            post_shift_host, r_dev, y_dev);
}

extern "C" int mvlm_conv2d_fast16(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                                  const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                                  const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev) {
    MVLM_ENTER(ctx);
    return conv2d_fast_impl(ctx, 2, x_dev, batch, cin, h, w, w_host, cout, bias_host, pre_scale_host, pre_shift_host, post_scale_host,
                            post_shift_host, r_dev, y_dev);
}
