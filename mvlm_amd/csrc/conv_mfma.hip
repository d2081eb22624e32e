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
#include <cstdlib>
#include <string>
#include <vector>

#include "common.h"
#include "conv_kernel.h"   // Cfg<> (tile geometry) for the compile-time queries below; kernels are not instantiated here
#include "conv_variants.h"
#include "conv_tuned.h"
#include "conv_tuned_net.h"
#include "conv_pair_tuned.h"
#include <map>
#include <mutex>
#include <algorithm>

// one launcher per variant, defined in conv_inst_g*.hip
#define X(id, name, ...)                                         \
    int mvlm_conv_launch_##id(mvlm_ctx* ctx, const ConvArgs& a); \
    int mvlm_conv_pair_launch_##id(mvlm_ctx* ctx, const ConvArgs& a0, const ConvArgs& a1);
MVLM_CONV_VARIANTS(X)
#undef X

namespace {

// level holds at most this many pixels in total -> latency-bound launch, split-K tiles
constexpr long SPLITK_MAX_PIXELS = 8192;

int pick_variant_rules(const ConvArgs& a) {
    if (a.ksize == 1) return (a.W >= 32 && a.cout_pad % 128 == 0) ? 4 : -1;
    if (a.ksize == 2) return (a.W >= 32 && a.H % 8 == 0) ? (a.cout_pad == 96 ? 11 : a.cout_pad == 80 ? 17 : a.cout_pad == 84 ? 29 : -1) : -1;
    if (a.cout_pad == 80) return (a.W >= 32 && a.H % 8 == 0) ? 16 : -1;  // 64 rows + one 16-row strip
    if (a.cout_pad % 32 == 20)  // 64 + 16 + 4 rows: plain layers writing a plain tensor (conv6 / conv10 of the 84-landmark net)
        return (a.cout_pad == 84 && a.W >= 32 && a.H % 8 == 0 && !a.amax_val && !a.up_out) ? 26 : -1;
    const long px = long(a.B) * a.H * a.W;
    // tiny feature maps / small batches: few pixels in total -> the split-K tiles (32 pixels per
    // workgroup, four waves share the K loop) shorten the serial chain; larger batches keep the
    // 128-pixel tiles
    const bool sk = a.cin_pad % 32 == 0 && px <= SPLITK_MAX_PIXELS && !a.amax_val;
    if (a.W == 32 && sk) return 15;
    if (a.W >= 32) {
        // two workgroups fit a CU: below ~512 workgroups the 128-pixel tiles fill the chip better
        if (a.cout_pad % 128 == 0) {
            const long blocks = px / 256 * (a.cout_pad / 128);
            return (blocks < 512 && !a.amax_val) ? 8 : 0;
        }
        if (a.cout_pad % 96 == 0) return 1;
        if (a.cout_pad % 64 == 0) {
            const long blocks = px / 256 * (a.cout_pad / 64);
            return blocks < 768 ? 9 : 10;
        }
        return (a.H % 16 == 0) ? 3 : -1;
    }
    if (a.W == 16) return sk ? 12 : (a.cin_pad % 16 == 0 ? 5 : -1);
    if (a.W == 8) return sk ? 13 : (a.cin_pad % 16 == 0 ? 6 : -1);
    if (a.W == 4) return sk ? 14 : (a.cin_pad % 16 == 0 ? 7 : -1);
    return -1;
}

// Measured choice first, the rules otherwise.  Order: a tuning override of this context (tools/tune_in_network.py), the
// in-network table (conv_tuned_net.h: every layer kind timed INSIDE a forward pass on real activations - a launch whose
// tensors come from HBM behind other layers ranks the tiles differently from one that re-reads them out of the MALL),
// the single-layer table (conv_tuned.h: mvlm_conv_bench on idle data, plain layers only).  The fused argmax / upsampling
// launches have one kernel that can serve them.
int pick_variant(const mvlm_ctx* ctx, const ConvArgs& a, bool rules_only = false) {
    const int by_rule = pick_variant_rules(a);
    if (rules_only || a.amax_val || a.up_in || a.up_out == 2 || a.ksize == 2 || a.H != a.W) return by_rule;
    const int kind = mvlm_conv_kind(a);
    if (ctx)
        for (const ConvOverride& o : ctx->conv_overrides)
            if (o.ksize == a.ksize && o.cin_pad == a.cin_pad && o.cout_pad == a.cout_pad && o.size == a.H && o.kind == kind) return o.variant;
    const short batch = short(a.B > 32767 ? 32767 : a.B);
    if (MVLM_CONV_TUNED_NET_N > 0) {
        const ConvTunedNet key = {short(a.ksize), short(a.cin_pad), short(a.cout_pad), short(a.H), short(kind), batch, 0};
        auto less = [](const ConvTunedNet& x, const ConvTunedNet& y) {
            if (x.ksize != y.ksize) return x.ksize < y.ksize;
            if (x.cin_pad != y.cin_pad) return x.cin_pad < y.cin_pad;
            if (x.cout_pad != y.cout_pad) return x.cout_pad < y.cout_pad;
            if (x.size != y.size) return x.size < y.size;
            if (x.kind != y.kind) return x.kind < y.kind;
            return x.batch < y.batch;
        };
        const ConvTunedNet* end = MVLM_CONV_TUNED_NET + MVLM_CONV_TUNED_NET_N;
        const ConvTunedNet* it = std::lower_bound(MVLM_CONV_TUNED_NET, end, key, less);  // smallest tuned batch >= B of this (shape, kind)
        if (it != end && it->ksize == key.ksize && it->cin_pad == key.cin_pad && it->cout_pad == key.cout_pad && it->size == key.size &&
            it->kind == key.kind)
            return it->variant;
    }
    if (MVLM_CONV_TUNED_N == 0) return by_rule;
    const ConvTuned key = {short(a.ksize), short(a.cin_pad), short(a.cout_pad), short(a.H), batch, 0};
    auto less = [](const ConvTuned& x, const ConvTuned& y) {
        if (x.ksize != y.ksize) return x.ksize < y.ksize;
        if (x.cin_pad != y.cin_pad) return x.cin_pad < y.cin_pad;
        if (x.cout_pad != y.cout_pad) return x.cout_pad < y.cout_pad;
        if (x.size != y.size) return x.size < y.size;
        return x.batch < y.batch;
    };
    const ConvTuned* end = MVLM_CONV_TUNED + MVLM_CONV_TUNED_N;
    const ConvTuned* it = std::lower_bound(MVLM_CONV_TUNED, end, key, less);  // smallest tuned batch >= B of this shape
    if (it == end || it->ksize != key.ksize || it->cin_pad != key.cin_pad || it->cout_pad != key.cout_pad || it->size != key.size)
        return by_rule;
    return it->variant;
}

}  // namespace

// Variant ids >= 256 are a split-K variant (low byte) whose input channels are divided over 2 / 4 / 8 workgroups per
// output tile (id = base + 256 log2(parts)); the partial tiles meet in a per-context workspace (one per launch stream).
int mvlm_conv_kparts_workspace(mvlm_ctx* ctx, float** ws, unsigned** cnt) {
    const int which = (ctx->launch_stream && ctx->launch_stream == ctx->cnn.side_stream) ? 1 : 0;
    if (!ctx->kparts_ws[0]) {  // both at once, outside any stream capture (a key's first pass is launch by launch)
        for (int i = 0; i < 2; ++i) {
            MVLM_CHECK_HIP(ctx, hipMalloc(&ctx->kparts_ws[i], size_t(MVLM_KPARTS_MAX_PARTS) * 1024 * sizeof(float)));
            MVLM_CHECK_HIP(ctx, hipMalloc(&ctx->kparts_cnt[i], size_t(MVLM_KPARTS_MAX_TILES) * sizeof(unsigned)));
            MVLM_CHECK_HIP(ctx, hipMemset(ctx->kparts_cnt[i], 0, size_t(MVLM_KPARTS_MAX_TILES) * sizeof(unsigned)));
        }
    }
    *ws = ctx->kparts_ws[which];
    *cnt = ctx->kparts_cnt[which];
    return 0;
}

const char* mvlm_conv_variant_name_impl(int v) {
    if (v == MVLM_CONV_VARIANT_FAST) return "conv3x3_bf16x3_t8x32";
    if (v == MVLM_CONV_VARIANT_FAST16) return "conv3x3_f16x2_t8x32";
    if (v >= 256) {
        // "<split-K variant>_k<parts>" and "<variant>_pair[_k<parts0>k<parts1>]": built on first use, kept for the process
        static std::mutex mu;
        static std::map<int, std::string> names;
        std::lock_guard<std::mutex> lk(mu);
        auto it = names.find(v);
        if (it == names.end()) {
            std::string n = mvlm_conv_variant_name_impl(v & 255);
            if (v & MVLM_CONV_PAIR_FLAG) {
                n += "_pair";
                const int l0 = (v >> 8) & 3, l1 = (v >> 10) & 3;
                if (l0 || l1) n += "_k" + std::to_string(1 << l0) + "k" + std::to_string(1 << l1);
            } else {
                n += "_k" + std::to_string(1 << ((v >> 8) & 3));
            }
            it = names.emplace(v, n).first;
        }
        return it->second.c_str();
    }
    switch (v) {
#define X(id, name, ...) \
    case id:             \
        return name;
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return "?";
}

// Would the dispatcher run this layer on the tile that can add a second, half-resolution input tensor on its load
// (ConvArgs::in2: the hourglass's "upsample x 2 + skip" on the consumer's side)?  The measured choice is respected: only where
// that tile is what the tables / rules pick anyway.
bool mvlm_conv_in2_ok(const mvlm_ctx* ctx, const ConvArgs& a) {
    if (a.ksize != 3 || a.up_in || a.up_out || a.amax_val || a.in_coff != 0 || (a.H & 1) || a.H != a.W) return false;
    if (ctx && ctx->conv_force_variant != -1) return false;
    return pick_variant(ctx, a) == 0;
}

int mvlm_conv_kind(const ConvArgs& a) { return a.up_out == 1 ? 1 : ((a.pool_out || a.pool_hint) ? 2 : 0); }

bool mvlm_conv_can_pool(const mvlm_ctx* ctx, const ConvArgs& a_in) {
    if (a_in.up_out || a_in.amax_val || (a_in.H & 1) || (a_in.W & 1)) return false;
    ConvArgs a = a_in;
    a.pool_hint = 1;
    if (ctx && ctx->conv_force_variant >= 0) return mvlm_conv_variant_can_pool(ctx->conv_force_variant);
    switch (pick_variant(ctx, a) & 255) {
#define X(id, name, ...)                                                                                   \
    case id: {                                                                                             \
        using V = __VA_ARGS__;                                                                             \
        return V::CAN_POOL_ANY;                                                                            \
    }
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return false;
}

bool mvlm_conv_variant_can_pool(int v) {
    switch (v & 255) {
#define X(id, name, ...)                                                                                   \
    case id: {                                                                                             \
        using V = __VA_ARGS__;                                                                             \
        return V::CAN_POOL_ANY;                                                                            \
    }
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return false;
}

namespace {
// can variant `v` (base id) serve this problem, and does it exist as a two-problem kernel?
bool pair_variant_serves(int v, const ConvArgs& a) {
    switch (v) {
#define X(id, name, ...)                                                                                                \
    case id: {                                                                                                          \
        using V = __VA_ARGS__;                                                                                          \
        return V::PAIRABLE && a.ksize == V::KS && a.W % V::TW == 0 && a.H % V::TRI == 0 && a.cout_pad % V::COUT_T == 0 && \
               a.cin_pad % V::CK == 0 && (!V::SPLITK || a.cin_pad % 32 == 0);                                           \
    }
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return false;
}
}  // namespace

// Should these two independent convolutions share a launch, and on which tiles?  Returns the pair's variant code
// (MVLM_CONV_PAIR_FLAG | base id | log2(kparts of problem 0) << 8 | log2(kparts of problem 1) << 10) or -1 (two launches).
//   mode 1: the measured table (conv_pair_tuned.h, tools/tune_conv_pairs.py: pairs that beat the two tuned single launches)
//   mode 2: always, on the tiles the dispatcher would give problem 0 when they can serve both (tests, tuning)
int mvlm_conv_pair_variant(const ConvArgs& a0, const ConvArgs& a1, int mode) {
    if (mode <= 0) return -1;
    for (const ConvArgs* a : {&a0, &a1})
        if (a->ksize != 3 || a->amax_val || a->up_in || a->up_out == 2 || a->n_par != 1 || a->H != a->W) return -1;
    if (a0.B != a1.B) return -1;
    int v = -1;
    if (mode == 2) {
        for (int cand : {pick_variant(nullptr, a0), pick_variant(nullptr, a1), pick_variant_rules(a0), pick_variant_rules(a1)}) {
            if (cand < 0) continue;
            const int base = cand & 255, lg = cand >> 8;
            if (pair_variant_serves(base, a0) && pair_variant_serves(base, a1)) {
                v = base | (lg << 8) | (lg << 10);
                break;
            }
        }
        return v < 0 ? -1 : (v | MVLM_CONV_PAIR_FLAG);
    }
    if (MVLM_CONV_PAIR_TUNED_N == 0 || a0.cin_pad != a1.cin_pad || a0.cout_pad != a1.cout_pad || a0.H != 2 * a1.H) return -1;
    const ConvPairTuned key = {short(a0.cin_pad), short(a0.cout_pad), short(a0.H), short(a0.B > 32767 ? 32767 : a0.B), 0};
    auto less = [](const ConvPairTuned& x, const ConvPairTuned& y) {
        if (x.cin_pad != y.cin_pad) return x.cin_pad < y.cin_pad;
        if (x.cout_pad != y.cout_pad) return x.cout_pad < y.cout_pad;
        if (x.size != y.size) return x.size < y.size;
        return x.batch < y.batch;
    };
    const ConvPairTuned* end = MVLM_CONV_PAIR_TUNED + MVLM_CONV_PAIR_TUNED_N;
    const ConvPairTuned* it = std::lower_bound(MVLM_CONV_PAIR_TUNED, end, key, less);  // smallest tuned batch >= B of this pair shape
    if (it == end || it->cin_pad != key.cin_pad || it->cout_pad != key.cout_pad || it->size != key.size || it->variant < 0) return -1;
    v = it->variant;
    if (!pair_variant_serves(v & 255, a0) || !pair_variant_serves(v & 255, a1)) return -1;
    // The table was measured without the pool kernel that follows a block whose tiles cannot emit the pooled tensor.  Since
    // round 5 every pairable tile but the one-row split-K tiles (t1x32) pools in its epilogue; for those, keep the pair only
    // if the single launch could not pool either (then the pool kernel runs in both forms and the comparison stands).
    for (const ConvArgs* a : {&a0, &a1})
        if (a->pool_hint && !mvlm_conv_variant_can_pool(v) && mvlm_conv_can_pool(nullptr, *a)) return -1;
    return v | MVLM_CONV_PAIR_FLAG;
}

int mvlm_launch_conv_pair(mvlm_ctx* ctx, const ConvArgs& a0, const ConvArgs& a1, int pair_variant) {
    MVLM_REQUIRE(ctx, pair_variant >= 0 && (pair_variant & MVLM_CONV_PAIR_FLAG) && pair_variant < 2 * MVLM_CONV_PAIR_FLAG, "conv: not a pair variant");
    ConvArgs b[2] = {a0, a1};
    for (int i = 0; i < 2; ++i) {
        const ConvArgs& a = b[i];
        MVLM_REQUIRE(ctx, a.in && a.w && a.B > 0 && a.H > 0 && a.W > 0 && a.H == a.W, "conv: null input / weights or bad shape");
        MVLM_REQUIRE(ctx, !a.up_in && a.up_out != 2 && (a.up_out != 1 || a.skip), "conv: a paired launch takes plain or scattering 3x3 convolutions");
        const double px = double(a.B) * a.H * a.W, lim = 4294967295.0;
        MVLM_REQUIRE(ctx, px * a.in_ctot < lim && (!a.out_raw || px * a.raw_ctot < lim) && (!a.res1 || px * a.res1_ctot < lim) &&
                              (!a.res2 || px * a.res2_ctot < lim) && (!a.out || px * a.out_ctot * (a.up_out ? 4 : 1) < lim) &&
                              (a.up_out != 1 || px * a.skip_ctot * 4 < lim) && (!a.pool_out || px / 4 * a.pool_ctot < lim),
                     "conv: a tensor exceeds 32-bit element offsets (lower the batch)");
        MVLM_REQUIRE(ctx, a.out || a.pool_out, "conv: no output requested");
        MVLM_REQUIRE(ctx, !a.pool_out || (mvlm_conv_variant_can_pool(pair_variant) && !(a.H & 1) && !a.up_out), "conv: this pair's kernel variant cannot emit the pooled tensor");
    }
    b[0].kparts = 1 << ((pair_variant >> 8) & 3);
    b[1].kparts = 1 << ((pair_variant >> 10) & 3);
    switch (pair_variant & 255) {
#define X(id, name, ...) \
    case id:             \
        return mvlm_conv_pair_launch_##id(ctx, b[0], b[1]);
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return ctx->fail("conv: unreachable variant");
}

int mvlm_conv_amax_parts(int H, int W) {
    // fused argmax is only built for the 8x32-pixel tiles (4 waves x one partial each)
    return (W / 32) * (H / 8) * 4;
}

int mvlm_launch_conv(mvlm_ctx* ctx, const ConvArgs& a, int* variant_out) {
    MVLM_REQUIRE(ctx, a.in && a.w && a.B > 0 && a.H > 0 && a.W > 0, "conv: null input / weights or empty shape");
    MVLM_REQUIRE(ctx, a.H == a.W, "conv: square feature maps only");
    MVLM_REQUIRE(ctx, !a.up_in || (a.H % 2 == 0), "conv: upsampled input needs even size");
    MVLM_REQUIRE(ctx, a.up_out != 1 || a.skip, "conv: up_out needs a skip tensor");
    MVLM_REQUIRE(ctx, a.ksize != 2 || ((a.sub_y | a.sub_x) & ~1) == 0, "conv: 2x2 window offset must be 0 or 1");
    const double px = double(a.B) * a.H * a.W, lim = 4294967295.0;
    MVLM_REQUIRE(ctx, px * a.in_ctot < lim, "conv: input tensor exceeds 32-bit element offsets (lower the batch)");
    MVLM_REQUIRE(ctx, !a.out_raw || px * a.raw_ctot < lim, "conv: raw output exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, !a.res1 || px * a.res1_ctot < lim, "conv: residual exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, !a.res2 || px * a.res2_ctot < lim, "conv: residual exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, !a.out || px * a.out_ctot * (a.up_out ? 4 : 1) < lim, "conv: output exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, a.up_out != 1 || px * a.skip_ctot * 4 < lim, "conv: skip tensor exceeds 32-bit element offsets");
    // conv_force_variant (mvlm_conv_bench only): >= 0 that variant, -2 the rules without the tuned table
    const int v = ctx->conv_force_variant >= 0 ? ctx->conv_force_variant : pick_variant(ctx, a, ctx->conv_force_variant == -2);
    MVLM_REQUIRE(ctx, v >= 0, "conv: no kernel variant for this shape");
    MVLM_REQUIRE(ctx, !a.in2 || v == 0, "conv: a second input tensor needs the 128-channel 8x32 tile (ask mvlm_conv_in2_ok first)");
    MVLM_REQUIRE(ctx, !a.in2 || px / 4 * a.in2_ctot < lim, "conv: second input exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, !a.pool_out || mvlm_conv_variant_can_pool(v), "conv: this shape's kernel variant cannot emit the pooled tensor");
    MVLM_REQUIRE(ctx, !a.pool_out || px / 4 * a.pool_ctot < lim, "conv: pooled output exceeds 32-bit element offsets");
    MVLM_REQUIRE(ctx, a.out || a.pool_out || a.amax_val, "conv: no output requested");
    if (variant_out) *variant_out = v;
    MVLM_REQUIRE(ctx, v < 1024, "conv: unknown kernel variant");
    ConvArgs b = a;
    b.kparts = 1 << (v >> 8);
    switch (v & 255) {
#define X(id, name, ...) \
    case id:             \
        return mvlm_conv_launch_##id(ctx, b);
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return ctx->fail("conv: unreachable variant");
}


// ---- tuning hooks (tools/tune_in_network.py) --------------------------------------------------------------------------------
// can kernel variant `variant` (>= 256: a split-K variant with its input channels over 2 / 4 workgroups) run a 3x3 layer of
// this shape and kind?  (kind 2: a tile that cannot pool in its epilogue is followed by the pool kernel; the 80- / 84-row tiles
// serve conv6 / conv10 only)
extern "C" int mvlm_conv_variant_serves(int variant, int ksize, int cin_pad, int cout_pad, int size, int kind) {
    if (variant < 0 || variant >= 1024 || ksize != 3 || kind < 0 || kind > 2) return 0;
    const int parts = 1 << (variant >> 8);
    switch (variant & 255) {
#define X(id, name, ...)                                                                                                     \
    case id: {                                                                                                               \
        using V = __VA_ARGS__;                                                                                               \
        if (V::KS != 3 || V::TAIL16 || V::COUT_T == 96) return 0;                                                            \
        if (size % V::TW != 0 || size % V::TRI != 0 || cout_pad % V::COUT_T != 0 || cin_pad % V::CK != 0) return 0;          \
        if (V::SPLITK && cin_pad % 32 != 0) return 0;                                                                        \
        if (parts > 1 && (!V::SPLITK || V::PIX_T != 32 || cin_pad % (parts * V::CK) != 0)) return 0;                         \
        return 1;                                                                                                            \
    }
        MVLM_CONV_VARIANTS(X)
#undef X
    }
    return 0;
}

// kernel variant for every launch of this (shape, kind) on this context, ahead of all tables; variant < 0 removes the
// entry, ksize == 0 removes all.  Captured launch graphs are dropped (they encode the kernels).
extern "C" int mvlm_conv_set_override(mvlm_ctx* ctx, int ksize, int cin_pad, int cout_pad, int size, int kind, int variant) {
    MVLM_ENTER(ctx);
    auto& ov = ctx->conv_overrides;
    if (ksize == 0) {
        ov.clear();
    } else {
        for (size_t i = 0; i < ov.size();)
            if (ov[i].ksize == ksize && ov[i].cin_pad == cin_pad && ov[i].cout_pad == cout_pad && ov[i].size == size && ov[i].kind == kind)
                ov.erase(ov.begin() + long(i));
            else
                ++i;
        if (variant >= 0) {
            MVLM_REQUIRE(ctx, mvlm_conv_variant_serves(variant, ksize, cin_pad, cout_pad, size, kind), "conv_set_override: the variant cannot serve this shape");
            ov.push_back({short(ksize), short(cin_pad), short(cout_pad), short(size), short(kind), variant});
        }
    }
    for (auto& g : ctx->cnn.graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    ctx->cnn.graphs.clear();
    return 0;
}
