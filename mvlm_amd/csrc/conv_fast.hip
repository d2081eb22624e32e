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
// Kernel: implicit GEMM like conv_mfma_kernel (same tile geometry 8 rows x 32 pixels, same accumulator layout, so the
// same kind of epilogue), one 512-thread workgroup per CU:
//   * 8 waves = 2 (output-channel halves) x 4 (pairs of pixel rows); a wave owns MT x 2 MFMA tiles (MT = COUT_T / 64)
//   * K runs in units of (16 input channels) x (one row of 3 taps).  LDS per stage:
//       X  [k-half 2][split 3][10 x 34 haloed pixels][8 channels] bf16     32 640 B   (restaged once per 3 units)
//       W  [tap 3][k-half 2][split 3][COUT_T][8 channels] bf16             288 * COUT_T B
//     both double-buffered (139 KB at COUT_T = 128): while unit u multiplies, the global loads of unit u + 1 are in
//     flight; they are converted / split and written to the other stage after the MFMAs; one barrier per unit.
//     A lane's MFMA fragment (8 consecutive channels of one pixel / one output channel) is one ds_read_b128, and
//     consecutive lanes read consecutive 16-byte slots: conflict-free.
//   * the consumer-side BatchNorm + ReLU is applied in fp32 before the split, zero padding after it.
//   * weights are split and laid out on the host ([chunk][tap row][tap][k-half][split][cout_pad][8], mvlm_pack_fast_weights)
//     so that staging W is a linear 16-byte copy.
// Layers it serves: 3x3, input channels a multiple of 16, output channels a multiple of 64, width a multiple of 32,
// height of 8, plain NCHW output with optional bias / post-BN+ReLU / raw copy / one residual.  Everything else (and
// every layer in the default "exact" precision) runs on conv_mfma_kernel.
#include <cstring>

#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

namespace {

constexpr int FT_TW = 32, FT_TRI = 8, FT_PW = FT_TW + 2, FT_PH = FT_TRI + 2, FT_NPIX = FT_PW * FT_PH;  // 340 haloed pixels
constexpr int FT_X_BYTES = 2 * 3 * FT_NPIX * 16;                                                      // 32 640
constexpr int FT_THREADS = 512;

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const __bf16 h = static_cast<__bf16>(x);  // round to nearest even (v_cvt_pk_bf16_f32)
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float bf16_value(unsigned short b) { return __uint_as_float(unsigned(b) << 16); }

// x -> (h, m, l) bf16 bit patterns with h + m + l == x up to 2^-24 |x|
__device__ __forceinline__ void split3(float x, unsigned short* h, unsigned short* m, unsigned short* l) {
    *h = bf16_bits(x);
    const float r1 = x - bf16_value(*h);
    *m = bf16_bits(r1);
    const float r2 = r1 - bf16_value(*m);
    *l = bf16_bits(r2);
}

template <int COUT_T>
__global__ __launch_bounds__(FT_THREADS, 2) void conv_bf16x3_kernel(const ConvArgs a, const unsigned short* __restrict__ wq,
                                                                   const int tiles_x, const int tiles_y, const int cout_tiles) {
    constexpr int MT = COUT_T / 64;                     // 32-row MFMA tiles per wave
    constexpr int W_BYTES = 3 * 2 * 3 * COUT_T * 16;    // one tap row
    constexpr int W_ITEMS = W_BYTES / 16;               // 16-byte items of a W stage
    constexpr int W_ITERS = (W_ITEMS + FT_THREADS - 1) / FT_THREADS;
    constexpr int X_ITEMS = 2 * FT_NPIX;                // (k-half, pixel) items of an X stage: 8 channels each
    constexpr int X_ITERS = (X_ITEMS + FT_THREADS - 1) / FT_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fast[];
    unsigned char* const sX = smem_fast;                        // 2 stages
    unsigned char* const sW = smem_fast + 2 * FT_X_BYTES;       // 2 stages
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
    const int x0 = tx * FT_TW, y0 = ty * FT_TRI, co0 = ct * COUT_T;
    const int H = a.H, W = a.W;
    const unsigned HW = unsigned(H) * unsigned(W);
    const int n_chunks = a.cin_pad / 16, n_units = 3 * n_chunks;

    // ---- staging plans ------------------------------------------------------------------------------------------
    // X: item e -> (k-half kh, haloed pixel p); eight channels 16 chunk + 8 kh + j of that pixel
    unsigned xoff[X_ITERS];   // element offset of channel (8 kh) of chunk 0 at that pixel, or ~0u outside the image
    int xdst[X_ITERS];        // byte offset inside an X stage of split 0 (+ FT_NPIX * 16 per split)
    int xkh[X_ITERS];
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
        const int e = tid + i * FT_THREADS;
        const int kh = e / FT_NPIX, p = e - kh * FT_NPIX;
        const int yy = p / FT_PW, xx = p - yy * FT_PW;
        const int y = y0 + yy - 1, x = x0 + xx - 1;
        const bool ok = e < X_ITEMS && y >= 0 && y < H && x >= 0 && x < W;
        xoff[i] = ok ? (unsigned(b0 * a.in_ctot + a.in_coff + 8 * kh) * HW + unsigned(y * W + x)) : 0xFFFFFFFFu;
        xdst[i] = e < X_ITEMS ? (kh * 3 * FT_NPIX + p) * 16 : -1;
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

    float xv[X_ITERS][8];
    u32x4 wv[W_ITERS];

    auto load_unit = [&](int u) __attribute__((always_inline)) {
        const int chunk = u / 3, row = u - chunk * 3;
        const unsigned short* const wb = wq + size_t(u) * w_block;
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i)
            wv[i] = *reinterpret_cast<const u32x4*>(wb + (wsrc[i] != 0xFFFFFFFFu ? wsrc[i] : 0u));
        if (row == 0) {
            const float* const base = a.in + size_t(chunk) * 16 * HW;
#pragma unroll
            for (int i = 0; i < X_ITERS; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = xoff[i] != 0xFFFFFFFFu && chunk * 16 + 8 * xkh[i] + j < a.cin;
                    xv[i][j] = base[ok ? xoff[i] + unsigned(j) * HW : 0u];
                }
        }
    };
    auto store_unit = [&](int u) __attribute__((always_inline)) {
        const int chunk = u / 3, row = u - chunk * 3;
        unsigned char* const wdst = sW + (u & 1) * W_BYTES;
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int f = tid + i * FT_THREADS;
            if (f < W_ITEMS) *reinterpret_cast<u32x4*>(wdst + f * 16) = wv[i];
        }
        if (row == 0) {
            unsigned char* const xd = sX + (chunk & 1) * FT_X_BYTES;
#pragma unroll
            for (int i = 0; i < X_ITERS; ++i) {
                if (xdst[i] < 0) continue;
                unsigned short hh[8], mm[8], ll[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = chunk * 16 + 8 * xkh[i] + j;
                    float v = xv[i][j];
                    if (a.pre_scale != nullptr) v = fmaxf(fmaf(v, sbn[c], sbn[256 + c]), 0.f);
                    const bool ok = xoff[i] != 0xFFFFFFFFu && c < a.cin;
                    v = ok ? v : 0.f;  // zero padding after the activation
                    split3(v, &hh[j], &mm[j], &ll[j]);
                }
                auto pack = [](const unsigned short (&s)[8]) {
                    u32x4 r;
                    r.x = unsigned(s[0]) | (unsigned(s[1]) << 16);
                    r.y = unsigned(s[2]) | (unsigned(s[3]) << 16);
                    r.z = unsigned(s[4]) | (unsigned(s[5]) << 16);
                    r.w = unsigned(s[6]) | (unsigned(s[7]) << 16);
                    return r;
                };
                *reinterpret_cast<u32x4*>(xd + xdst[i]) = pack(hh);
                *reinterpret_cast<u32x4*>(xd + xdst[i] + FT_NPIX * 16) = pack(mm);
                *reinterpret_cast<u32x4*>(xd + xdst[i] + 2 * FT_NPIX * 16) = pack(ll);
            }
        }
    };

    // ---- accumulators and per-lane LDS offsets ----------------------------------------------------------------------
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // B fragment of pixel row n of this wave, tap (dy, dx): haloed pixel ((2 wn + n + dy) * PW + l31 + dx), k-half = half
    int boff[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) boff[n] = (half * 3 * FT_NPIX + (2 * wn + n) * FT_PW + l31) * 16;
    // A fragment of MFMA tile m: output channel wm * 32 MT + 32 m + l31, k-half = half
    int aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = (half * 3 * COUT_T + wm * 32 * MT + 32 * m + l31) * 16;

    // ---- prologue ---------------------------------------------------------------------------------------------------
    if (a.pre_scale != nullptr)
        for (int i = tid; i < a.cin_pad; i += FT_THREADS) {
            sbn[i] = a.pre_scale[i];
            sbn[256 + i] = a.pre_shift[i];
        }
    load_unit(0);
    __syncthreads();  // BatchNorm table visible
    store_unit(0);
    __syncthreads();

    for (int u = 0; u < n_units; ++u) {
        const bool more = u + 1 < n_units;
        if (more) load_unit(u + 1);
        const int chunk = u / 3, row = u - chunk * 3;
        const unsigned char* const xs = sX + (chunk & 1) * FT_X_BYTES;
        const unsigned char* const ws = sW + (u & 1) * W_BYTES;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int poff = (row * FT_PW + t) * 16;  // tap (row, t): dy = row, dx = t in the haloed tile
            bf16x8 af[MT][3], bf[2][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m][s] = *reinterpret_cast<const bf16x8*>(ws + ((t * 6 + s) * COUT_T) * 16 + aoff[m]);
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    bf[n][s] = *reinterpret_cast<const bf16x8*>(xs + s * FT_NPIX * 16 + boff[n] + poff);
            }
            // six cross terms, smallest first: (w, x) = (h,l) (l,h) (m,m) | (h,m) (m,h) | (h,h)
            constexpr int WS[6] = {0, 2, 1, 0, 1, 0}, XS[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][WS[p]], bf[n][XS[p]], acc[m][n], 0, 0, 0);
        }
        if (more) store_unit(u + 1);
        __syncthreads();
    }

    // ---- epilogue: bias, post-BatchNorm + ReLU, raw copy, residual, store -----------------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 * MT + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co >= a.cout) continue;
            const float bias = a.bias ? a.bias[co] : 0.f;
            const float ps = a.post_scale ? a.post_scale[co] : 1.f, pt = a.post_scale ? a.post_shift[co] : 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = y0 + 2 * wn + n, x = x0 + l31;
                const unsigned pix = unsigned(y * W + x);
                float v = acc[m][n][r] + bias;
                if (a.post_scale) v = fmaxf(fmaf(v, ps, pt), 0.f);
                if (a.out_raw) a.out_raw[(size_t(b0) * a.raw_ctot + a.raw_coff + co) * HW + pix] = v;
                if (a.res1) v += a.res1[(size_t(b0) * a.res1_ctot + a.res1_coff + co) * HW + pix];
                if (a.out) a.out[(size_t(b0) * a.out_ctot + a.out_coff + co) * HW + pix] = v;
            }
        }
}

template <int COUT_T>
int launch_fast(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq) {
    constexpr size_t LDS = size_t(2) * FT_X_BYTES + size_t(2) * (3 * 2 * 3 * COUT_T * 16) + 2 * 256 * sizeof(float);
    static_assert(LDS <= 160 * 1024, "stages must fit the CU's LDS");
    const int bit = COUT_T == 128 ? 62 : 63;
    if (!((ctx->conv_attr_mask >> bit) & 1ull)) {
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_kernel<COUT_T>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS)));
        ctx->conv_attr_mask |= 1ull << bit;
    }
    const int tiles_x = a.W / FT_TW, tiles_y = a.H / FT_TRI, cout_tiles = a.cout_pad / COUT_T;
    const long nblk = long(tiles_x) * tiles_y * a.B * cout_tiles;
    MVLM_REQUIRE(ctx, nblk > 0 && nblk < (1l << 31), "conv_fast: bad grid");
    hipLaunchKernelGGL((conv_bf16x3_kernel<COUT_T>), dim3((unsigned)nblk), dim3(FT_THREADS), LDS, ctx->cur_stream(), a, wq,
                       tiles_x, tiles_y, cout_tiles);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace

// can the fast kernel serve this launch?
bool mvlm_conv_fast_ok(const ConvArgs& a) {
    return a.ksize == 3 && a.cin_pad % 16 == 0 && a.cin_pad <= 256 && a.cout_pad % 64 == 0 && a.W % FT_TW == 0 && a.H % FT_TRI == 0 &&
           !a.up_in && !a.up_out && !a.res2 && !a.skip && !a.amax_val && !a.pool_out && (a.out || a.out_raw);
}

int mvlm_launch_conv_fast(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq_dev) {
    MVLM_REQUIRE(ctx, mvlm_conv_fast_ok(a) && wq_dev, "conv_fast: launch not eligible");
    const double px = double(a.B) * a.H * a.W, lim = 4294967295.0;
    MVLM_REQUIRE(ctx, px * a.in_ctot < lim, "conv_fast: input tensor exceeds 32-bit element offsets (lower the batch)");
    return a.cout_pad % 128 == 0 ? launch_fast<128>(ctx, a, wq_dev) : launch_fast<64>(ctx, a, wq_dev);
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
    const int cin_pad = (cin + 15) / 16 * 16, cout_pad = (cout + 63) / 64 * 64;
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
