// Implicit-GEMM 3x3 / 1x1 convolution on the CDNA4 matrix cores, exact fp32.
//
// This is the kernel the landmark network (reference: MVLMModel,
// src/mvlm/prediction/paulsenpredictor.py:364-432) spends ~99 % of its time in.
//
// Mapping (gfx950, wave64):
//   GEMM  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],   k = (tap, cin)
//   v_mfma_f32_32x32x2_f32: A = weights (lane l: cout = l&31, k = l>>5),
//                           B = activations (lane l: pixel = l&31, k = l>>5),
//   so an accumulator register holds 32 consecutive pixels of one output channel per
//   half-wave -> 128-byte coalesced stores into planar NCHW tensors.
//   The MFMA is a k-ordered fp32 fma chain (bit-exact fp32, no reduced precision), which
//   is what the 1e-3 landmark parity bound needs.
//
//   A workgroup (4 waves) owns COUT_T output channels x PIX_T pixels.  Per K-chunk of 8
//   input channels it stages into LDS
//     sX [8][NIMG][TRI+2][TW+2]  the haloed input tile, with the consumer's BatchNorm+ReLU
//                                (pre-activation blocks, paulsenpredictor.py:269-271)
//                                applied on the way in and the zero padding inserted
//                                AFTER the activation, optionally read through a nearest
//                                2x upsample (paulsenpredictor.py:428-429),
//     sW [taps][8][COUT_T]       the weight slice (host-packed [tap][cin][cout]),
//   then issues taps*4 k-steps of MT x NT MFMAs straight out of LDS (conflict-free:
//   32 consecutive floats per half-wave for both operands).
//   Several workgroups are resident per CU (<= 48 KB LDS, <= 168 VGPRs), so one group's
//   staging overlaps another's MFMA stream.
//
//   The epilogue fuses bias, post-BatchNorm+ReLU, the residual-block concat/add
//   (paulsenpredictor.py:273), the hourglass "upsample + skip add" (:334-359) as a 2x2
//   scatter, and for the last layer the per-(view, landmark) argmax
//   (paulsenpredictor.py:123) so the [N,NL,256,256] heatmaps never reach HBM.
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int CK = 8;  // input channels per LDS stage
constexpr unsigned INVALID_OFF = 0xFFFFFFFFu;

template <int COUT_T_, int TW_, int TRI_, int NIMG_, int KS_>
struct Cfg {
    static constexpr int COUT_T = COUT_T_, TW = TW_, TRI = TRI_, NIMG = NIMG_, KS = KS_;
    static constexpr int TAPS = KS * KS;
    static constexpr int HALO = KS / 2;
    static constexpr int PW = TW + 2 * HALO;
    static constexpr int PH = TRI + 2 * HALO;
    static constexpr int PLANE = NIMG * PH * PW;  // floats per channel in sX
    static constexpr int PIX_T = TW * TRI * NIMG;
    static constexpr int XT = CK * PLANE;
    static constexpr int XT_PAD = (XT + 3) / 4 * 4;
    static constexpr int WT = TAPS * CK * COUT_T;
    static constexpr int MT = COUT_T / 32;
    static constexpr int NT = PIX_T / 4 / 32;
    static constexpr int X_ITERS = (XT + 255) / 256;
    static constexpr int W_ITERS = (WT / 4 + 255) / 256;
    static constexpr size_t LDS_BYTES = size_t(XT_PAD + WT) * 4;
    static_assert(PIX_T % 128 == 0, "pixel tile must split into 4 waves x 32-pixel MFMA columns");
    static_assert(COUT_T % 32 == 0, "cout tile must be a multiple of the 32-row MFMA tile");
};

template <class C, bool AMAX>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                        const int cout_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;
    float* sW = smem + C::XT_PAD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order: consecutive block ids land on different XCDs, so give every
    // XCD a contiguous run of tiles (cout tiles of one pixel tile share its input in L2).
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ct = lid % cout_tiles;
    const int pt = lid / cout_tiles;
    const int tx = pt % tiles_x;
    const int ty = (pt / tiles_x) % tiles_y;
    const int tb = pt / (tiles_x * tiles_y);
    const int x0 = tx * C::TW, y0 = ty * C::TRI, b0 = tb * C::NIMG, co0 = ct * C::COUT_T;

    const int H = a.H, W = a.W;
    const int Hin = a.up_in ? (H >> 1) : H, Win = a.up_in ? (W >> 1) : W;
    const unsigned HWin = unsigned(Hin) * unsigned(Win);

    // ---- per-thread staging plan for the input tile (fixed across K-chunks) ----------
    unsigned goff[C::X_ITERS];
#pragma unroll
    for (int i = 0; i < C::X_ITERS; ++i) {
        const int e = tid + i * 256;
        const int c = e / C::PLANE;
        const int rem = e - c * C::PLANE;
        const int img = rem / (C::PH * C::PW);
        const int rem2 = rem - img * (C::PH * C::PW);
        const int yy = rem2 / C::PW;
        const int xx = rem2 - yy * C::PW;
        const int y = y0 + yy - C::HALO, x = x0 + xx - C::HALO, b = b0 + img;
        const bool ok = (e < C::XT) && y >= 0 && y < H && x >= 0 && x < W && b < a.B;
        const int ys = a.up_in ? (y >> 1) : y, xs = a.up_in ? (x >> 1) : x;
        goff[i] = ok ? (unsigned(b * a.in_ctot + a.in_coff + c) * HWin + unsigned(ys * Win + xs)) : INVALID_OFF;
    }

    // ---- per-lane pixel offsets inside sX for the NT pixel columns of this wave ------
    int pixoff[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int p = wave * (C::PIX_T / 4) + n * 32 + l31;
        const int x = p % C::TW;
        const int rr = p / C::TW;
        const int yl = rr % C::TRI;
        const int img = rr / C::TRI;
        pixoff[n] = (img * C::PH + yl) * C::PW + x + half * C::PLANE;
    }
    const int woff = half * C::COUT_T + l31;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const bool has_pre = a.pre_scale != nullptr;

    for (int cb = 0; cb < a.cin_pad; cb += CK) {
        // -------- stage input tile: global -> (BN+ReLU) -> LDS ------------------------
        float xv[C::X_ITERS];
#pragma unroll
        for (int i = 0; i < C::X_ITERS; ++i) {
            const int c = cb + (tid + i * 256) / C::PLANE;
            const bool ok = goff[i] != INVALID_OFF && c < a.cin;
            xv[i] = ok ? a.in[size_t(goff[i]) + size_t(cb) * HWin] : 0.f;
        }
        // -------- stage weight slice --------------------------------------------------
        float4 wv[C::W_ITERS];
#pragma unroll
        for (int i = 0; i < C::W_ITERS; ++i) {
            const int f = tid + i * 256;
            if (f < C::WT / 4) {
                const int row = f / (C::COUT_T / 4);
                const int c4 = f - row * (C::COUT_T / 4);
                const int tap = row / CK;
                const int c = row - tap * CK;
                wv[i] = *reinterpret_cast<const float4*>(a.w + (size_t(tap) * a.cin_pad + cb + c) * a.cout_pad + co0 +
                                                         c4 * 4);
            }
        }
        if (has_pre) {
#pragma unroll
            for (int i = 0; i < C::X_ITERS; ++i) {
                const int c = cb + (tid + i * 256) / C::PLANE;
                const bool ok = goff[i] != INVALID_OFF && c < a.cin;
                if (ok) {
                    const float s = a.pre_scale[c], t = a.pre_shift[c];
                    xv[i] = fmaxf(fmaf(xv[i], s, t), 0.f);
                }
            }
        }
        __syncthreads();  // previous chunk's MFMA reads are done
#pragma unroll
        for (int i = 0; i < C::X_ITERS; ++i) {
            const int e = tid + i * 256;
            if (e < C::XT) sX[e] = xv[i];
        }
#pragma unroll
        for (int i = 0; i < C::W_ITERS; ++i) {
            const int f = tid + i * 256;
            if (f < C::WT / 4) reinterpret_cast<float4*>(sW)[f] = wv[i];
        }
        __syncthreads();

        // -------- taps x channel pairs of MFMAs --------------------------------------
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            const int toff = (tap / C::KS) * C::PW + (tap % C::KS);
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                float av[C::MT], bv[C::NT];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) av[m] = sW[(tap * CK + 2 * cp) * C::COUT_T + m * 32 + woff];
#pragma unroll
                for (int n = 0; n < C::NT; ++n) bv[n] = sX[2 * cp * C::PLANE + pixoff[n] + toff];
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---------------------------------- epilogue ---------------------------------------
    const size_t HW = size_t(H) * W;
    // per pixel column of this wave: image, flat pixel index, validity
    int pb[C::NT], ppix[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int p = wave * (C::PIX_T / 4) + n * 32 + l31;
        const int rr = p / C::TW;
        const int b = b0 + rr / C::TRI;
        pb[n] = b < a.B ? b : -1;
        ppix[n] = (y0 + rr % C::TRI) * W + x0 + p % C::TW;
    }
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool co_ok = co < a.cout;
            const int coc = co_ok ? co : 0;
            const float bias = a.bias ? a.bias[coc] : 0.f;
            const float ps = a.post_scale ? a.post_scale[coc] : 1.f;
            const float pt = a.post_scale ? a.post_shift[coc] : 0.f;
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                if (co_ok && pb[n] >= 0) {
                    const size_t b = size_t(pb[n]);
                    const size_t pix = size_t(ppix[n]);
                    float v = acc[m][n][r];
                    if (a.bias) v += bias;
                    if (a.post_scale) v = fmaxf(fmaf(v, ps, pt), 0.f);
                    if (a.out_raw) a.out_raw[(b * a.raw_ctot + a.raw_coff + co) * HW + pix] = v;
                    if (a.res1) {
                        float rv = a.res1[(b * a.res1_ctot + a.res1_coff + co) * HW + pix];
                        if (a.res2) rv += a.res2[(b * a.res2_ctot + a.res2_coff + co) * HW + pix];
                        v += rv;
                    }
                    if (a.out) {
                        if (!a.up_out) {
                            a.out[(b * a.out_ctot + a.out_coff + co) * HW + pix] = v;
                        } else {
                            const int y = ppix[n] / W, x = ppix[n] - y * W;
                            const size_t W2 = size_t(W) * 2;
                            const size_t o2 = size_t(2 * y) * W2 + 2 * x;
                            const size_t obase = (b * a.out_ctot + a.out_coff + co) * HW * 4 + o2;
                            const size_t sbase = (b * a.skip_ctot + a.skip_coff + co) * HW * 4 + o2;
                            const float2 s0 = *reinterpret_cast<const float2*>(a.skip + sbase);
                            const float2 s1 = *reinterpret_cast<const float2*>(a.skip + sbase + W2);
                            *reinterpret_cast<float2*>(a.out + obase) = make_float2(v + s0.x, v + s0.y);
                            *reinterpret_cast<float2*>(a.out + obase + W2) = make_float2(v + s1.x, v + s1.y);
                        }
                    }
                }
            }
        }
    }

    if constexpr (AMAX) {
        // Fused heatmap argmax (paulsenpredictor.py:123): conv11 has no residual / post-BN, so
        // the heatmap value is acc + bias.  First maximum in row-major order wins ties.
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int coc = co < a.cout ? co : 0;
                const float bias = a.bias ? a.bias[coc] : 0.f;
                float best_v = -INFINITY;
                int best_i = 0x7fffffff;
#pragma unroll
                for (int n = 0; n < C::NT; ++n) {
                    const float v = acc[m][n][r] + bias;
                    const bool better = pb[n] >= 0 && (v > best_v || (v == best_v && ppix[n] < best_i));
                    best_v = better ? v : best_v;
                    best_i = better ? ppix[n] : best_i;
                }
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1) {
                    const float ov = __shfl_xor(best_v, s);
                    const int oi = __shfl_xor(best_i, s);
                    const bool better = ov > best_v || (ov == best_v && oi < best_i);
                    best_v = better ? ov : best_v;
                    best_i = better ? oi : best_i;
                }
                if (l31 == 0 && co < a.cout && b0 < a.B) {
                    const size_t o = (size_t(b0) * a.cout + co) * a.amax_parts + (size_t(ty) * tiles_x + tx) * 4 + wave;
                    a.amax_val[o] = best_v;
                    a.amax_idx[o] = best_i;
                }
            }
        }
    }
}

// ---- variant table ---------------------------------------------------------------------
// id, name, instantiation.  W >= 32 uses row-segment tiles; smaller levels fold rows /
// images into the 32-pixel MFMA column.
#define MVLM_CONV_VARIANTS(X)                                  \
    X(0, "conv3x3_c128_t8x32", Cfg<128, 32, 8, 1, 3>)          \
    X(1, "conv3x3_c96_t8x32", Cfg<96, 32, 8, 1, 3>)            \
    X(2, "conv3x3_c64_t16x32", Cfg<64, 32, 16, 1, 3>)          \
    X(3, "conv3x3_c32_t16x32", Cfg<32, 32, 16, 1, 3>)          \
    X(4, "conv1x1_c128_t8x32", Cfg<128, 32, 8, 1, 1>)          \
    X(5, "conv3x3_c128_t16x16", Cfg<128, 16, 16, 1, 3>)        \
    X(6, "conv3x3_c64_t16x16", Cfg<64, 16, 16, 1, 3>)          \
    X(7, "conv3x3_c128_t8x8x4", Cfg<128, 8, 8, 4, 3>)          \
    X(8, "conv3x3_c64_t8x8x4", Cfg<64, 8, 8, 4, 3>)            \
    X(9, "conv3x3_c128_t4x4x16", Cfg<128, 4, 4, 16, 3>)        \
    X(10, "conv3x3_c64_t4x4x16", Cfg<64, 4, 4, 16, 3>)

template <class C>
int launch_variant(mvlm_ctx* ctx, const ConvArgs& a) {
    const int tiles_x = a.W / C::TW, tiles_y = a.H / C::TRI;
    const int tiles_b = (a.B + C::NIMG - 1) / C::NIMG;
    const int cout_tiles = a.cout_pad / C::COUT_T;
    MVLM_REQUIRE(ctx, a.W % C::TW == 0 && a.H % C::TRI == 0, "conv: spatial size not a multiple of the tile");
    MVLM_REQUIRE(ctx, a.cout_pad % C::COUT_T == 0, "conv: cout_pad not a multiple of the cout tile");
    MVLM_REQUIRE(ctx, a.cin_pad % CK == 0, "conv: cin_pad must be a multiple of 8");
    if (a.amax_val) {
        MVLM_REQUIRE(ctx, C::NIMG == 1, "conv: fused argmax needs one image per tile");
        MVLM_REQUIRE(ctx, a.amax_parts == tiles_x * tiles_y * 4, "conv: argmax partial count mismatch");
        MVLM_REQUIRE(ctx, !a.res1 && !a.post_scale && !a.up_out, "conv: fused argmax expects a plain conv + bias layer");
    }
    const long nblk = long(tiles_x) * tiles_y * tiles_b * cout_tiles;
    MVLM_REQUIRE(ctx, nblk > 0 && nblk < (1l << 31), "conv: bad grid");
    if (a.amax_val) {
        if constexpr (C::NIMG == 1 && C::TW == 32 && C::TRI == 8 && C::KS == 3) {
            hipLaunchKernelGGL((conv_mfma_kernel<C, true>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, ctx->stream,
                               a, tiles_x, tiles_y, cout_tiles);
        } else {
            return ctx->fail("conv: fused argmax is only built for the 8x32-pixel tile variants");
        }
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<C, false>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, ctx->stream, a,
                           tiles_x, tiles_y, cout_tiles);
    }
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

int pick_variant(const ConvArgs& a) {
    if (a.ksize == 1) return (a.W >= 32 && a.cout_pad % 128 == 0) ? 4 : -1;
    const bool c128 = a.cout_pad % 128 == 0;
    const bool c64 = a.cout_pad % 64 == 0;
    if (a.W >= 32) {
        if (c128) return 0;
        if (a.cout_pad % 96 == 0) return 1;
        if (c64) return (a.H % 16 == 0) ? 2 : -1;
        return (a.H % 16 == 0) ? 3 : -1;
    }
    if (a.W == 16) return c128 ? 5 : (c64 ? 6 : -1);
    if (a.W == 8) return c128 ? 7 : (c64 ? 8 : -1);
    if (a.W == 4) return c128 ? 9 : (c64 ? 10 : -1);
    return -1;
}

}  // namespace

const char* mvlm_conv_variant_name_impl(int v) {
    switch (v) {
#define X(id, name, ...) \
    case id:             \
        return name;
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return "?";
}

int mvlm_conv_amax_parts(int H, int W) {
    // fused argmax is only used with variants 0/1 (8x32 tiles, 4 waves)
    return (W / 32) * (H / 8) * 4;
}

int mvlm_launch_conv(mvlm_ctx* ctx, const ConvArgs& a, int* variant_out) {
    MVLM_REQUIRE(ctx, a.in && a.w && a.B > 0 && a.H > 0 && a.W > 0, "conv: null input / weights or empty shape");
    MVLM_REQUIRE(ctx, a.H == a.W, "conv: square feature maps only");
    MVLM_REQUIRE(ctx, !a.up_in || (a.H % 2 == 0), "conv: upsampled input needs even size");
    MVLM_REQUIRE(ctx, !a.up_out || a.skip, "conv: up_out needs a skip tensor");
    const double in_elems = double(a.B) * a.in_ctot * a.H * a.W;
    MVLM_REQUIRE(ctx, in_elems < 4.0e9, "conv: input tensor exceeds 32-bit element offsets");
    const int v = pick_variant(a);
    MVLM_REQUIRE(ctx, v >= 0, "conv: no kernel variant for this shape");
    if (variant_out) *variant_out = v;
    switch (v) {
#define X(id, name, ...) \
    case id:             \
        return launch_variant<__VA_ARGS__>(ctx, a);
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return ctx->fail("conv: unreachable variant");
}
