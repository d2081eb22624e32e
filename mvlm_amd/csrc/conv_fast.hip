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
// Layers it serves: 3x3, input channels a multiple of 16, output channels a multiple of 64, width a multiple of 32,
// height of 8, plain NCHW output with optional bias / post-BN+ReLU / raw copy / one residual.  Everything else (and
// every layer in the default "exact" precision) runs on conv_mfma_kernel.
#include <cstring>
#include <type_traits>

#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

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

__device__ __forceinline__ unsigned bf16_bits(float x) {
    const __bf16 h = static_cast<__bf16>(x);  // round to nearest even (v_cvt_pk_bf16_f32)
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float bf16_value(unsigned b) { return __uint_as_float(b << 16); }

// x -> (h, m, l) bf16 bit patterns with h + m + l == x up to 2^-24 |x|
__device__ __forceinline__ void split3(float x, unsigned* h, unsigned* m, unsigned* l) {
    *h = bf16_bits(x);
    const float r1 = x - bf16_value(*h);
    *m = bf16_bits(r1);
    const float r2 = r1 - bf16_value(*m);
    *l = bf16_bits(r2);
}

// COUT_T output channels x (TRI rows x 32 pixels); 8 waves = 2 channel halves x 4 groups of TRI / 4 rows
template <int COUT_T, int TRI>
struct FastCfg {
    static constexpr int MT = COUT_T / 64;               // 32-row MFMA tiles per wave (channels)
    static constexpr int NT = TRI / 4;                    // 32-pixel row segments per wave
    static constexpr int PH = TRI + 2, NPIX = FT_PW * PH;  // haloed tile
    static constexpr int X_BYTES = 2 * 3 * NPIX * 16;      // [k-half][split][pixel][8 ch]
    static constexpr int W_BYTES = 3 * 2 * 3 * COUT_T * 16;  // one tap row: [tap][k-half][split][cout][8 ch]
    static constexpr int W_ITEMS = W_BYTES / 16, W_ITERS = (W_ITEMS + FT_THREADS - 1) / FT_THREADS;
    static constexpr int X_ITEMS = 2 * NPIX, X_ITERS = (X_ITEMS + FT_THREADS - 1) / FT_THREADS;
    static constexpr int TAP_MFMAS = 6 * MT * NT;          // per tap and wave
    static constexpr size_t LDS = size_t(2) * X_BYTES + size_t(2) * W_BYTES + 2 * 256 * sizeof(float);
    static_assert(LDS <= 160 * 1024, "stages must fit the CU's LDS");
    static_assert(TAP_MFMAS == 24, "the staging schedule below is written for 24 MFMAs per tap");
    static_assert(X_ITERS <= 3 && W_ITERS <= 5, "staging schedule");
};

// GEN: the rarely needed parts (input channels that do not fill the last chunk, a second residual) are compiled in
template <int COUT_T, int TRI, bool GEN>
__global__ __launch_bounds__(FT_THREADS, 2) void conv_bf16x3_kernel(const ConvArgs a, const unsigned short* __restrict__ wq,
                                                                   const int tiles_x, const int tiles_y, const int cout_tiles) {
    using C = FastCfg<COUT_T, TRI>;
    constexpr int MT = C::MT, NT = C::NT, NPIX = C::NPIX, X_BYTES = C::X_BYTES, W_BYTES = C::W_BYTES;
    constexpr int W_ITEMS = C::W_ITEMS, W_ITERS = C::W_ITERS, X_ITEMS = C::X_ITEMS, X_ITERS = C::X_ITERS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fast[];
    unsigned char* const sX = smem_fast;                        // 2 stages
    unsigned char* const sW = smem_fast + 2 * X_BYTES;          // 2 stages
    float* const sbn = reinterpret_cast<float*>(sW + 2 * W_BYTES);  // [2][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

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
    unsigned xoff[X_ITERS];   // element offset of channel (8 kh) of chunk 0 at that pixel, or ~0u outside the image
    int xdst[X_ITERS];        // byte offset inside an X stage of split 0 (+ NPIX * 16 per split), < 0: no item
    int xkh[X_ITERS];
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
        const int e = tid + i * FT_THREADS;
        const int kh = e / NPIX, p = e - kh * NPIX;
        const int yy = p / FT_PW, xx = p - yy * FT_PW;
        const int y = y0 + yy - 1, x = x0 + xx - 1;
        const bool ok = e < X_ITEMS && y >= 0 && y < H && x >= 0 && x < W;
        xoff[i] = ok ? (unsigned(b0 * a.in_ctot + a.in_coff + 8 * kh) * HW + unsigned(y * W + x)) : 0xFFFFFFFFu;
        xdst[i] = e < X_ITEMS ? (kh * 3 * NPIX + p) * 16 : -1;
        xkh[i] = kh;
    }
    // W: item f -> segment (tap, kh, split) and output channel c of the tile; linear in the host layout
    unsigned wsrc[W_ITERS];   // u16 element offset inside a (chunk, tap row) block of the packed weights
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
        const int f = tid + i * FT_THREADS;
        const int seg = f / COUT_T, c = f - seg * COUT_T;
        wsrc[i] = f < W_ITEMS ? unsigned((seg * a.cout_pad + co0 + c) * 8) : 0xFFFFFFFFu;
    }
    const size_t w_block = size_t(18) * a.cout_pad * 8;  // u16 elements of one (chunk, tap row) block

    float xv[X_ITERS][8];      // the next chunk's activations, in flight / waiting for their split
    unsigned xq[3][4];         // one item's eight channels as bf16 pairs, per split
    u32x4 wv[W_ITERS];         // the next tap row's weights

    // ---- staging micro-operations (each one small enough for the shadow of one or two MFMAs) --------------------------
    auto load_w = [&](int unit) __attribute__((always_inline)) {
        const unsigned short* const wb = wq + size_t(unit) * w_block;
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i)
            wv[i] = *reinterpret_cast<const u32x4*>(wb + (wsrc[i] != 0xFFFFFFFFu ? wsrc[i] : 0u));
    };
    auto store_w = [&](auto ic, int unit) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int f = tid + i * FT_THREADS;
        if (f < W_ITEMS) *reinterpret_cast<u32x4*>(sW + (unit & 1) * W_BYTES + f * 16) = wv[i];
    };
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        const float* const base = a.in + size_t(chunk) * 16 * HW;
#pragma unroll
        for (int i = 0; i < X_ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = xoff[i] != 0xFFFFFFFFu && (!partial_cin || chunk * 16 + 8 * xkh[i] + j < a.cin);
                xv[i][j] = base[ok ? xoff[i] + unsigned(j) * HW : 0u];  // branch-free; what is not there is zeroed in convert_pair
            }
    };
    // BatchNorm + ReLU, zero padding, split: channels 2 jp and 2 jp + 1 of item i -> one u32 per split
    auto convert_pair = [&](auto ic, auto jc, int chunk) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, jp = decltype(jc)::value;
        unsigned h[2], m[2], l[2];
        const int c = chunk * 16 + 8 * xkh[i] + 2 * jp;
        float2 sc = make_float2(1.f, 1.f), sh = make_float2(0.f, 0.f);
        if (has_bn) {
            sc = *reinterpret_cast<const float2*>(sbn + c);
            sh = *reinterpret_cast<const float2*>(sbn + 256 + c);
        }
        const bool inside = xoff[i] != 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = xv[i][2 * jp + k];
            if (has_bn) v = fmaxf(fmaf(v, k ? sc.y : sc.x, k ? sh.y : sh.x), 0.f);
            v = inside && (!partial_cin || c + k < a.cin) ? v : 0.f;  // zero padding after the activation; channels past cin
            split3(v, &h[k], &m[k], &l[k]);
        }
        xq[0][jp] = h[0] | (h[1] << 16);
        xq[1][jp] = m[0] | (m[1] << 16);
        xq[2][jp] = l[0] | (l[1] << 16);
    };
    auto store_x = [&](auto ic, auto sc, int chunk) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, sp = decltype(sc)::value;
        if (xdst[i] >= 0) {
            u32x4 r;
            r.x = xq[sp][0];
            r.y = xq[sp][1];
            r.z = xq[sp][2];
            r.w = xq[sp][3];
            *reinterpret_cast<u32x4*>(sX + (chunk & 1) * X_BYTES + xdst[i] + sp * NPIX * 16) = r;
        }
    };
    // all of item i at once (prologue)
    auto stage_item = [&](auto ic, int chunk) __attribute__((always_inline)) {
        static_for<0, 4>([&](auto jc) { convert_pair(ic, jc, chunk); });
        static_for<0, 3>([&](auto sc) { store_x(ic, sc, chunk); });
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
    for (int n = 0; n < NT; ++n) boff[n] = (half * 3 * NPIX + (NT * wn + n) * FT_PW + l31) * 16;
    // A fragment of MFMA tile m: output channel wm * 32 MT + 32 m + l31, k-half = half
    int aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = (half * 3 * COUT_T + wm * 32 * MT + 32 * m + l31) * 16;

    // ---- prologue: BatchNorm table, chunk 0's activations, unit 0's weights ----------------------------------------------
    if (has_bn)
        for (int i = tid; i < a.cin_pad; i += FT_THREADS) {  // padding channels: relu(0 * 0 + 0) = 0
            sbn[i] = i < a.cin ? a.pre_scale[i] : 0.f;
            sbn[256 + i] = i < a.cin ? a.pre_shift[i] : 0.f;
        }
    load_w(0);
    load_x(0);
    __syncthreads();  // BatchNorm table visible
    static_for<0, W_ITERS>([&](auto ic) { store_w(ic, 0); });
    static_for<0, X_ITERS>([&](auto ic) { stage_item(ic, 0); });
    __syncthreads();

    // One unit = (chunk, tap row ROW): 3 taps x 24 MFMAs per wave.  The staging of what comes next is cut into
    // micro-operations placed BETWEEN the MFMAs (slot q = 24 tap + index): the matrix pipe holds a wave's vector issue
    // for 8 of an MFMA's 32 cycles, the rest is where BatchNorm / split / LDS writes of the next stage run.
    //   every unit : global loads of the next unit's weights at q = 0, their LDS writes at q = 24 + 4 i (tap 1)
    //   ROW == 0   : global loads of the NEXT chunk's activations at q = 1 (two units of lead)
    //   ROW == 1   : item 0 of the next chunk: pairs at q = 2, 8, 14, 20, splits stored at q = 44, 50, 56
    //   ROW == 2   : item 1: pairs at q = 2, 6, 10, 14, stores at q = 18, 20, 22; item 2: pairs at q = 28, 32, 36, 40,
    //                stores at q = 60, 64, 68
    auto unit = [&](auto rowc, int chunk) __attribute__((always_inline)) {
        constexpr int ROW = decltype(rowc)::value;
        const int u = 3 * chunk + ROW;
        const bool more_w = u + 1 < 3 * n_chunks, more_x = chunk + 1 < n_chunks;
        const unsigned char* const xs = sX + (chunk & 1) * X_BYTES;
        const unsigned char* const ws = sW + (u & 1) * W_BYTES;
        bf16x8 af[MT][3], bf[NT][3];
        static_for<0, 3>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int poff = (ROW * FT_PW + t) * 16;  // tap (ROW, t): dy = ROW, dx = t in the haloed tile
#if defined(MVLM_FAST_ABLATE_ONE_TAP)  // timing experiment only: fragments are read for the first tap of a unit only
            if constexpr (t == 0)
#endif
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m][sp] = *reinterpret_cast<const bf16x8*>(ws + ((t * 6 + sp) * COUT_T) * 16 + aoff[m]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bf[n][sp] = *reinterpret_cast<const bf16x8*>(xs + sp * NPIX * 16 + boff[n] + poff);
            }
            // six cross terms, smallest first: (w, x) = (h,l) (l,h) (m,m) | (h,m) (m,h) | (h,h)
            constexpr int WS[6] = {0, 2, 1, 0, 1, 0}, XS[6] = {2, 0, 1, 1, 0, 0};
            static_for<0, 24>([&](auto ic) {
                constexpr int i = decltype(ic)::value, q = 24 * t + i;
                constexpr int p = i / (MT * NT), mn = i % (MT * NT), m = mn / NT, n = mn % NT;
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][WS[p]], bf[n][XS[p]], acc[m][n], 0, 0, 0);
                // ---- side work of slot q ----
#if defined(MVLM_FAST_ABLATE_NO_STAGING)  // timing experiment only: wrong results
                if constexpr (q < 0) {
#else
                if constexpr (q >= 0) {
#endif
                if constexpr (q == 0) {
                    if (more_w) load_w(u + 1);
                }
                if constexpr (q == 1 && ROW == 0) {
                    if (more_x) load_x(chunk + 1);
                }
                if constexpr (q >= 24 && q < 24 + 4 * W_ITERS && (q - 24) % 4 == 0) {
                    if (more_w) store_w(std::integral_constant<int, (q - 24) / 4>{}, u + 1);
                }
                if constexpr (ROW == 1) {
                    if constexpr (q == 2 || q == 8 || q == 14 || q == 20) {
                        if (more_x) convert_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, (q - 2) / 6>{}, chunk + 1);
                    }
                    if constexpr (q == 44 || q == 50 || q == 56) {
                        if (more_x) store_x(std::integral_constant<int, 0>{}, std::integral_constant<int, (q - 44) / 6>{}, chunk + 1);
                    }
                }
                if constexpr (ROW == 2 && X_ITERS >= 2) {
                    if constexpr (q == 2 || q == 6 || q == 10 || q == 14) {
                        if (more_x) convert_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, (q - 2) / 4>{}, chunk + 1);
                    }
                    if constexpr (q == 18 || q == 20 || q == 22) {
                        if (more_x) store_x(std::integral_constant<int, 1>{}, std::integral_constant<int, (q - 18) / 2>{}, chunk + 1);
                    }
                }
                if constexpr (ROW == 2 && X_ITERS >= 3) {
                    if constexpr (q == 28 || q == 32 || q == 36 || q == 40) {
                        if (more_x) convert_pair(std::integral_constant<int, 2>{}, std::integral_constant<int, (q - 28) / 4>{}, chunk + 1);
                    }
                    if constexpr (q == 60 || q == 64 || q == 68) {
                        if (more_x) store_x(std::integral_constant<int, 2>{}, std::integral_constant<int, (q - 60) / 4>{}, chunk + 1);
                    }
                }
                }
            });
        });
        __syncthreads();
    };
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        unit(std::integral_constant<int, 0>{}, chunk);
        unit(std::integral_constant<int, 1>{}, chunk);
        unit(std::integral_constant<int, 2>{}, chunk);
    }

    // ---- epilogue: bias, post-BatchNorm + ReLU, raw copy, residual, store -----------------------------------------------
    // One MFMA tile row (32 channels x NT row segments) at a time: all its residual values are requested first, then
    // the stores follow; the layer kinds that carry the time run without per-element feature tests.
    auto epilogue = [&](auto raw_c, auto res_c, auto par_c) __attribute__((always_inline)) {
        constexpr bool RAW = decltype(raw_c)::value, RES = decltype(res_c)::value;
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cob + (r & 3) + 8 * (r >> 2);
                if (co >= a.cout) continue;
                float bias = 0.f, ps = 1.f, pt = 0.f;
                if (PAR == 1 || PAR == 2 || (PAR == 3 && a.bias)) bias = a.bias[co];
                const bool post = PAR == 2 || (PAR == 3 && a.post_scale);
                if (post) {
                    ps = a.post_scale[co];
                    pt = a.post_shift[co];
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float v = acc[m][n][r] + bias;
                    if (post) v = fmaxf(fmaf(v, ps, pt), 0.f);
                    if (RAW) a.out_raw[(size_t(b0) * a.raw_ctot + a.raw_coff + co) * HW + pix[n]] = v;
                    if constexpr (RES) v += resv[r][n];
                    if (GEN && PAR == 3 && a.res2) v += a.res2[(size_t(b0) * a.res2_ctot + a.res2_coff + co) * HW + pix[n]];
                    if (!RAW || a.out) a.out[(size_t(b0) * a.out_ctot + a.out_coff + co) * HW + pix[n]] = v;
                }
            }
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    const bool has_raw = a.out_raw != nullptr, has_res = a.res1 != nullptr;
    const bool plain = !a.bias && !a.post_scale && !a.res2;
    if (has_raw && has_res && plain && a.out)
        epilogue(T_{}, T_{}, std::integral_constant<int, 0>{});   // residual block conv1 / conv2
    else if (!has_raw && has_res && plain)
        epilogue(F_{}, T_{}, std::integral_constant<int, 0>{});   // residual block conv3
    else if (!has_raw && !has_res && a.bias && a.post_scale && !a.res2)
        epilogue(F_{}, F_{}, std::integral_constant<int, 2>{});   // conv5 / conv9
    else if (has_raw && has_res)
        epilogue(T_{}, T_{}, std::integral_constant<int, 3>{});
    else if (has_raw)
        epilogue(T_{}, F_{}, std::integral_constant<int, 3>{});
    else if (has_res)
        epilogue(F_{}, T_{}, std::integral_constant<int, 3>{});
    else
        epilogue(F_{}, F_{}, std::integral_constant<int, 3>{});
}

template <int COUT_T, int TRI, bool GEN>
int launch_fast(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq) {
    using C = FastCfg<COUT_T, TRI>;
    const int bit = 60 + (COUT_T == 128 ? 0 : 1) + (GEN ? 2 : 0);
    if (!((ctx->conv_attr_mask >> bit) & 1ull)) {
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_kernel<COUT_T, TRI, GEN>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS)));
        ctx->conv_attr_mask |= 1ull << bit;
    }
    const int tiles_x = a.W / FT_TW, tiles_y = a.H / TRI, cout_tiles = a.cout_pad / COUT_T;
    const long nblk = long(tiles_x) * tiles_y * a.B * cout_tiles;
    MVLM_REQUIRE(ctx, nblk > 0 && nblk < (1l << 31), "conv_fast: bad grid");
    hipLaunchKernelGGL((conv_bf16x3_kernel<COUT_T, TRI, GEN>), dim3((unsigned)nblk), dim3(FT_THREADS), C::LDS, ctx->cur_stream(), a, wq,
                       tiles_x, tiles_y, cout_tiles);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace

// can the fast kernel serve this launch?  (its own paddings: mvlm_fast_cin_pad / mvlm_fast_cout_pad, common.h)
bool mvlm_conv_fast_ok(const ConvArgs& a) {
    if (a.ksize != 3 || !mvlm_fast_channels_ok(a.cin, a.cout)) return false;
    // 128-channel tiles cover 8 rows, 64-channel tiles 16 rows (same matrix work per staged input tile)
    const int rows = mvlm_fast_cout_pad(a.cout) % 128 == 0 ? 8 : 16;
    return a.W % FT_TW == 0 && a.H % rows == 0 && !a.up_in && !a.up_out && !a.skip && !a.amax_val && !a.pool_out && (a.out || a.out_raw);
}

int mvlm_launch_conv_fast(mvlm_ctx* ctx, const ConvArgs& exact, const unsigned short* wq_dev) {
    MVLM_REQUIRE(ctx, mvlm_conv_fast_ok(exact) && wq_dev, "conv_fast: launch not eligible");
    ConvArgs a = exact;  // the split weights carry their own paddings
    a.cin_pad = mvlm_fast_cin_pad(a.cin);
    a.cout_pad = mvlm_fast_cout_pad(a.cout);
    const double px = double(a.B) * a.H * a.W, lim = 4294967295.0;
    MVLM_REQUIRE(ctx, px * a.in_ctot < lim, "conv_fast: input tensor exceeds 32-bit element offsets (lower the batch)");
    if (a.cin != a.cin_pad || a.res2)
        return a.cout_pad % 128 == 0 ? launch_fast<128, 8, true>(ctx, a, wq_dev) : launch_fast<64, 16, true>(ctx, a, wq_dev);
    return a.cout_pad % 128 == 0 ? launch_fast<128, 8, false>(ctx, a, wq_dev) : launch_fast<64, 16, false>(ctx, a, wq_dev);
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

// ---- single convolution through the fast kernel (test hook, mirrors mvlm_conv2d) ---------------------------------------
extern "C" int mvlm_conv2d_fast(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                                const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                                const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, x_dev && w_host && y_dev, "conv2d_fast: null pointer");
    MVLM_REQUIRE(ctx, mvlm_fast_channels_ok(cin, cout), "conv2d_fast: channel counts outside what the fast kernel serves");
    const int cin_pad = mvlm_fast_cin_pad(cin), cout_pad = mvlm_fast_cout_pad(cout);
    std::vector<uint16_t> wq(mvlm_pack_fast_weights(w_host, cout, cin, cout_pad, cin_pad, nullptr));
    MVLM_REQUIRE(ctx, !wq.empty() && mvlm_pack_fast_weights(w_host, cout, cin, cout_pad, cin_pad, wq.data()) == wq.size(),
                 "conv2d_fast: weight packing failed");
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
    if (mvlm_launch_conv_fast(ctx, a, dq)) return 1;
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
