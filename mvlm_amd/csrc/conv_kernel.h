// Kernel template of the implicit-GEMM convolution (see conv_mfma.hip for the overview); included by
// the translation units that instantiate groups of variants (conv_inst_g*.hip).
#ifndef MVLM_CONV_KERNEL_H
#define MVLM_CONV_KERNEL_H
#include <cstdlib>
#include <string>
#include <type_traits>

#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr unsigned INVALID_OFF = 0xFFFFFFFFu;

// compile-time loop: the body sees its index as a constant, so register arrays indexed by it
// can never be demoted to scratch by an unrolling heuristic
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// COUT_T x (TW x TRI x NIMG) output tile, KS x KS taps, CK input channels per LDS stage
// (PIX_T == 64, round 4: TWO 32-pixel columns per workgroup on one staged weight slice - a launch of the small levels is
//  bound by staging 36 bytes of weights per input channel and output row for 32 pixels of matrix work; a second column
//  halves that traffic per MFMA.  These tiles have no K-parts form and no fused argmax.)
// SPLITK: the four waves of a workgroup share ONE 32x32 output tile and each sums a quarter of
// every K-chunk's channels (latency-bound tiny feature maps: 4x more workgroups, 4x shorter serial
// K loop per wave); the partial sums are added in wave order through LDS, i.e. deterministically.
template <int COUT_T_, int TW_, int TRI_, int NIMG_, int KS_, int CK_, bool SPLITK_ = false>
struct Cfg {
    static constexpr int COUT_T = COUT_T_, TW = TW_, TRI = TRI_, NIMG = NIMG_, KS = KS_, CK = CK_;
    static constexpr bool SPLITK = SPLITK_;
    static constexpr int CKW = SPLITK ? CK / 4 : CK;  // channels of a chunk one wave multiplies
    static constexpr int TAPS = KS * KS;
    static constexpr int HALO = KS == 1 ? 0 : 1;  // KS == 2: a 2x2 window inside the 3x3 halo tile
    static constexpr int PW = TW + 2 * HALO;
    static constexpr int PH = TRI + 2 * HALO;
    static constexpr int PLANE = NIMG * PH * PW;  // floats per channel in sX
    static constexpr int PIX_T = TW * TRI * NIMG;
    static constexpr int XT = CK * PLANE;
    static constexpr int XT_PAD = (XT + 3) / 4 * 4;
    static constexpr int WT = TAPS * CK * COUT_T;
    static constexpr int STAGE = XT_PAD + WT;  // floats per LDS stage (two stages)
    static constexpr int MT = COUT_T / 32;  // full 32-row MFMA tiles
    // COUT_T = 32*MT + 16: the last 16 output channels run on v_mfma_f32_16x16x4_f32 (same FLOP rate,
    // half the rows), so a 73-landmark layer pads to 80 rows instead of 96
    // COUT_T = 32*MT + 16 + 4: a further 4-row strip on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 channels x 4 pixels per
    // instruction, one k each) - 84 landmarks fit exactly instead of padding to 96
    static constexpr bool TAIL4 = COUT_T % 32 == 20;
    static constexpr bool TAIL16 = COUT_T % 32 == 16 || TAIL4;
    // the fused argmax serves the last layer only (73 / 84 landmarks -> 80 / 96-row tiles of 8x32 pixels)
    static constexpr bool HAS_AMAX = NIMG == 1 && TW == 32 && TRI == 8 && KS != 1 && (COUT_T == 80 || COUT_T == 96 || (COUT_T == 84 && KS == 2));
    static constexpr int NT = SPLITK ? PIX_T / 32 : PIX_T / 4 / 32;  // split-K: every wave multiplies all (one or two) 32-pixel columns
    static constexpr int NT16 = TAIL16 ? 2 * NT : 1;  // 16-pixel column groups of a wave
    static constexpr int KSTEPS = TAPS * CKW / 2;
    static constexpr int X_ITERS = (XT + 255) / 256;
    static constexpr int W_ITERS = (WT / 4 + 255) / 256;
    static constexpr int BN_MAXC = 256;  // pre-BN scale/shift of up to 256 input channels live in LDS
    static constexpr size_t LDS_BYTES = size_t(2 * STAGE + 2 * BN_MAXC) * 4;
    // accumulators + staged tile + operands: above ~200 registers the kernel is told it owns
    // the whole SIMD register file (one wave per SIMD) instead of spilling for occupancy
    static constexpr int ACC_REGS = (COUT_T / 32) * (SPLITK ? PIX_T / 32 : TW * TRI * NIMG / 128) * 16 + (TAIL16 ? 4 * NT16 : 0) + (TAIL4 ? 4 : 0);
    // register budget per lane: 168 at three workgroups per CU, 256 at two
    // (four per CU = 128 registers makes the 64-accumulator tiles spill; measured slower)
    static constexpr int MIN_BLOCKS_PER_CU = (ACC_REGS <= 64 && TW * TRI * NIMG <= 256) ? 3 : 2;
    // fused 2x2 max-pool in the epilogue.  POOL32: the 32-pixel-row tiles - the rows of a pair are two pixel registers of a
    // lane, the columns two neighbouring lanes.  POOL_SMALL (round 5): tiles narrower than 32 pixels (and the split-K tiles) -
    // a 32-pixel MFMA column then holds 32 / TW consecutive rows of ONE image, so the row partner of lane l is lane l ^ TW and
    // the column partner lane l ^ 1; TRI even keeps the pairs inside the tile.  max is exact: the pooled tensor is bit for bit
    // what the pool kernel makes of the full-resolution one.
    static constexpr bool POOL32 = !SPLITK && TW == 32 && NIMG == 1 && (PIX_T / 4 / 32) % 2 == 0 && COUT_T % 32 == 0;
    static constexpr bool POOL_SMALL = TW < 32 && TRI % 2 == 0 && COUT_T % 32 == 0;
    static constexpr bool CAN_POOL_ANY = POOL32 || POOL_SMALL;
    // the tile that also exists with a second input tensor added on the load (ConvArgs::in2): the dominant 128 x (8 x 32) tile
    static constexpr bool HAS_IN2 = !SPLITK && COUT_T == 128 && TW == 32 && TRI == 8 && NIMG == 1 && KS == 3 && CK == 4;
    // variants that also exist as a two-problem launch (conv_pair_kernel): the tiles of the residual blocks' 3x3 convolutions
    static constexpr bool PAIRABLE = KS == 3 && COUT_T % 32 == 0 && COUT_T != 96 && !(SPLITK && PIX_T != 32);
    static_assert(SPLITK ? ((PIX_T == 32 || PIX_T == 64) && COUT_T == 32 && CK % 8 == 0) : (PIX_T % 128 == 0),
                  "pixel tile must split into 4 waves x 32-pixel MFMA columns (or be one or two columns for split-K)");
    static_assert(!TAIL4 || PIX_T == 256, "the 4-row strip gives every lane of a wave one pixel: 64 pixels per wave");
    static_assert(COUT_T % 32 == 0 || (TAIL16 && !SPLITK && CK == 4 && TW == 32 && NIMG == 1),
                  "cout tile must be a multiple of the 32-row MFMA tile (+ one 16-row strip on the 32-pixel-row tiles)");
    static_assert(CK % 2 == 0, "the f32 MFMA consumes two k values per step");
    static_assert(LDS_BYTES <= 160 * 1024, "two stages must fit the CU's 160 KiB LDS");
};

// ---- pieces of the main loop, as force-inlined functions over register arrays -----------
// One "item" is one staged element per thread: items [0, X_ITERS) are input-tile floats,
// items [X_ITERS, X_ITERS + W_ITERS) are float4s of the weight slice.
template <class C>
struct StageRegs {
    float xv[C::X_ITERS];
    float xv2[C::X_ITERS];  // IN2 kernels only: the low-resolution tensor's value of the same element
    f32x4 wv[C::W_ITERS];
    float bn_s[C::X_ITERS], bn_t[C::X_ITERS];
};

// 4-row strip (TAIL4): lane l supplies the weight of channel 32 MT + 16 + (l & 3) (A row of every 4x4 block) and the
// input of ITS pixel (B column l & 3 of block l >> 2); accumulator register v = channel 32 MT + 16 + v at the lane's pixel
struct Strip4 {
    int woff4 = 0, pixoff4 = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
};

// Split-K tiles: what a lane keeps from the start of the kernel for its share of the epilogue (see conv_mfma_kernel)
struct SplitKTail {
    bool ok = false;
    unsigned b = 0, pix = 0;
    int y = 0, x = 0;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
};
template <int N>
struct SplitKTails {
    SplitKTail t[N];
};

// Issue the global load of item T for K-chunk cb.  Loads are unconditional (out-of-tile /
// padding elements read element 0 and are zeroed at write time): a branch around a load makes
// the compiler wait for each one separately.
template <class C, int T, bool BN_FROM_GLOBAL = false, bool IN2 = false>
__device__ __forceinline__ void issue_item(const ConvArgs& a, int cb, int tid, unsigned HWin, const float* sbn,
                                           const unsigned (&goff)[C::X_ITERS], const unsigned (&woff_g)[C::W_ITERS],
                                           StageRegs<C>& r, float* st_dst, const unsigned (&goff2)[C::X_ITERS]) {
    if constexpr (T < C::X_ITERS) {
        const int c = cb + (tid + T * 256) / C::PLANE;
        const bool ok = goff[T] != INVALID_OFF && c < a.cin;
        // wave-uniform base; one unconditional load per lane (masked lanes read a valid element)
        const float* const base = a.in + size_t(cb < a.cin ? cb : 0) * HWin;
        r.xv[T] = base[ok ? goff[T] : 0u];
        if constexpr (IN2) {  // the same element of the half-resolution tensor (four staged elements share one: L1 / L2 hits)
            const float* const base2 = a.in2 + size_t(cb < a.cin ? cb : 0) * (HWin >> 2);
            r.xv2[T] = base2[ok ? goff2[T] : 0u];
        }
        if (a.pre_scale != nullptr) {  // this element's BatchNorm scale / shift from the LDS copy
            const int cc = c < a.cin_pad ? c : 0;
            if constexpr (BN_FROM_GLOBAL) {  // first chunk: the LDS copy is still being filled
                r.bn_s[T] = a.pre_scale[cc];
                r.bn_t[T] = a.pre_shift[cc];
            } else {
                r.bn_s[T] = sbn[cc];
                r.bn_t[T] = sbn[C::BN_MAXC + cc];
            }
        }
    } else {
        constexpr int I = T - C::X_ITERS;
        const float* const base = a.w + size_t(cb) * a.cout_pad;  // wave-uniform
        r.wv[I] = *reinterpret_cast<const f32x4*>(base + (woff_g[I] != INVALID_OFF ? woff_g[I] : 0u));
    }
}

// BatchNorm + ReLU (pre-activation block), zero padding AFTER the activation, then the LDS write
template <class C, int T, bool IN2 = false>
__device__ __forceinline__ void write_item(const ConvArgs& a, int cb, int tid, float* st,
                                           const unsigned (&goff)[C::X_ITERS], const StageRegs<C>& r) {
    if constexpr (T < C::X_ITERS) {
        const int e = tid + T * 256;
        const int c = cb + e / C::PLANE;
        const bool ok = goff[T] != INVALID_OFF && c < a.cin;
        float v = r.xv[T];
        if constexpr (IN2) v += r.xv2[T];  // skip + upsampled low: the sum the scatter form leaves in the skip tensor, bit for bit
        if (a.pre_scale != nullptr) v = fmaxf(fmaf(v, r.bn_s[T], r.bn_t[T]), 0.f);
        v = ok ? v : 0.f;
        if (e < C::XT) st[e] = v;
    } else {
        constexpr int I = T - C::X_ITERS;
        const int f = tid + I * 256;
        if (f < C::WT / 4) reinterpret_cast<f32x4*>(st + C::XT_PAD)[f] = r.wv[I];
    }
}

// The MFMAs of one K-chunk out of LDS stage `st`.  Operands of k-step s+1 are fetched from LDS
// while the MFMAs of step s issue.  With STAGE_NEXT the staging of the next chunk (cb_next) is
// spread over the k-steps so its address arithmetic, loads, BatchNorm and LDS writes run in the
// shadow of the 64-cycle MFMAs: loads are issued during the first half of the steps, written to
// the other stage `st_next` half a chunk (~9k cycles) later.
// TR: the accumulators hold the TRANSPOSED tile (rows = pixels, columns = output channels): the operands of
// every MFMA are swapped, nothing else changes.  The fused-argmax launches use it: a lane then owns ONE output
// channel and 16 pixels per tile, so the per-channel maximum is a chain of in-register compares.
template <class C, bool STAGE_NEXT, bool TR = false, bool IN2 = false>
__device__ __forceinline__ void compute_chunk(const ConvArgs& a, const float* st, float* st_next, int cb_next, int tid,
                                              unsigned HWin, const float* sbn, int woff, const int (&pixoff)[C::NT],
                                              const unsigned (&goff)[C::X_ITERS], const unsigned (&woff_g)[C::W_ITERS],
                                              StageRegs<C>& r, f32x16 (&acc)[C::MT][C::NT], int woff16,
                                              const int (&pixoff16)[C::NT16], f32x4 (&acc16)[C::NT16], Strip4& s4,
                                              const unsigned (&goff2)[C::X_ITERS]) {
    constexpr int T_TOT = C::X_ITERS + C::W_ITERS;
    // staging schedule of the next chunk: loads issued over the first ISSUE_SPAN k-steps, LDS writes over the last WRITE_SPAN
#if !defined(MVLM_STAGE_ISSUE_DIV)
#define MVLM_STAGE_ISSUE_DIV 2
#endif
#if !defined(MVLM_STAGE_WRITE_DIV)
#define MVLM_STAGE_WRITE_DIV 2
#endif
    constexpr int ISSUE_SPAN = C::KSTEPS / MVLM_STAGE_ISSUE_DIV > 0 ? C::KSTEPS / MVLM_STAGE_ISSUE_DIV : 1;
    constexpr int WRITE_SPAN = C::KSTEPS / MVLM_STAGE_WRITE_DIV > 0 ? C::KSTEPS / MVLM_STAGE_WRITE_DIV : 1;
    constexpr int WRITE_START = C::KSTEPS - WRITE_SPAN;
    float av[2][C::MT], bv[2][C::NT];
    float a16 = 0.f, b16[C::NT16];  // 16-row strip: operands of one tap's four channels (two k-steps)
    float a4[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, b4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // 4-row strip: [k-step parity][channel of the pair]
    if constexpr (C::TAIL4) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a4[0][c] = st[s4.woff4 + c * C::COUT_T];
            b4[0][c] = st[s4.pixoff4 + c * C::PLANE];
        }
    }
#pragma unroll
    for (int m = 0; m < C::MT; ++m) av[0][m] = st[woff + m * 32];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) bv[0][n] = st[pixoff[n]];
    static_for<0, C::KSTEPS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int nx = ks + 1;
        if constexpr (nx < C::KSTEPS) {
            constexpr int tap = nx / (C::CKW / 2), cp = nx % (C::CKW / 2);
            constexpr int toff = (tap / C::KS) * C::PW + (tap % C::KS);
#pragma unroll
            for (int m = 0; m < C::MT; ++m) av[nx & 1][m] = st[woff + (tap * C::CK + 2 * cp) * C::COUT_T + m * 32];
#pragma unroll
            for (int n = 0; n < C::NT; ++n) bv[nx & 1][n] = st[2 * cp * C::PLANE + pixoff[n] + toff];
        }
        if constexpr (C::TAIL4 && nx < C::KSTEPS) {
            constexpr int tap4 = nx / (C::CKW / 2), cp4 = nx % (C::CKW / 2);
            constexpr int toff4 = (tap4 / C::KS) * C::PW + (tap4 % C::KS);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                a4[nx & 1][c] = st[s4.woff4 + (tap4 * C::CK + 2 * cp4 + c) * C::COUT_T];
                b4[nx & 1][c] = st[s4.pixoff4 + (2 * cp4 + c) * C::PLANE + toff4];
            }
        }
        if constexpr (C::TAIL16 && (ks & 1) == 0) {
            // the strip's A (16 channels x 4 k) and B (4 k x 16 pixels) fragments of this tap, used
            // by the 16x16x4 MFMAs of the tap's second k-step
            constexpr int tap16 = ks / 2;
            constexpr int toff16 = (tap16 / C::KS) * C::PW + (tap16 % C::KS);
            a16 = st[woff16 + tap16 * C::CK * C::COUT_T];
#pragma unroll
            for (int j = 0; j < C::NT16; ++j) b16[j] = st[pixoff16[j] + toff16];
        }
        // two MFMAs first, then this step's share of the staging work, then the rest: the side
        // work's LDS / VMEM operations complete in the shadow of the remaining MFMAs instead of
        // being waited for at the next step's lgkmcnt(0)
#if !defined(MVLM_MFMA_LEAD)
#define MVLM_MFMA_LEAD 2
#endif
        constexpr int LEAD = (C::MT * C::NT >= 4) ? (MVLM_MFMA_LEAD < C::MT * C::NT ? MVLM_MFMA_LEAD : C::MT * C::NT) : 1;
        static_for<0, C::MT * C::NT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int m = i / C::NT, n = i % C::NT;
            if constexpr (i < LEAD)
                acc[m][n] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[ks & 1][n], av[ks & 1][m], acc[m][n], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][n], acc[m][n], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGE_NEXT) {
            static_for<0, T_TOT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
#if !defined(MVLM_ABLATE_NO_LOADS)
                if constexpr ((t * ISSUE_SPAN) / T_TOT == ks) issue_item<C, t, false, IN2>(a, cb_next, tid, HWin, sbn, goff, woff_g, r, st_next, goff2);
#endif
#if !defined(MVLM_ABLATE_NO_WRITES)
                if constexpr (WRITE_START + (t * WRITE_SPAN) / T_TOT == ks) write_item<C, t, IN2>(a, cb_next, tid, st_next, goff, r);
#endif
            });
            __builtin_amdgcn_sched_barrier(0);
        }
        // two waves share a SIMD (two workgroups per CU): the one inside its MFMA run keeps the pipe
        __builtin_amdgcn_s_setprio(1);
        static_for<0, C::MT * C::NT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int m = i / C::NT, n = i % C::NT;
            if constexpr (i >= LEAD)
                acc[m][n] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[ks & 1][n], av[ks & 1][m], acc[m][n], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][n], acc[m][n], 0, 0, 0);
        });
        if constexpr (C::TAIL16 && (ks & 1) == 1) {
#pragma unroll
            for (int j = 0; j < C::NT16; ++j)
                acc16[j] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(b16[j], a16, acc16[j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x4f32(a16, b16[j], acc16[j], 0, 0, 0);
        }
        if constexpr (C::TAIL4) {
            // (the strip has one form: with TR the 32-row and 16-row tiles are transposed around it, register v of the
            //  strip stays channel 32 MT + 16 + v at the lane's pixel)
#pragma unroll
            for (int c = 0; c < 2; ++c) s4.acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[ks & 1][c], b4[ks & 1][c], s4.acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        // keep each step's LDS prefetch and side work inside its own MFMA shadow
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Software-pipelined main loop, one workgroup (4 waves, one per SIMD) per CU-resident tile:
//   while the MFMAs of K-chunk c run out of LDS stage c&1, the global loads of chunk c+1 are
//   in flight into registers; after the MFMAs they get their BatchNorm+ReLU and are written to
//   the other stage; ONE barrier per chunk.  The kernel may use the whole 512-register file
//   (launch bounds 256,1), so nothing spills and the 128 accumulators stay in registers.
//
// conv_tile is the whole workgroup program; the kernels below only say which problem a workgroup belongs to:
// conv_mfma_kernel = one convolution per launch (block `bid` of `nblk`), conv_pair_kernel = two independent convolutions
// of the same tile configuration in one grid.
template <class C, bool AMAX, bool IN2 = false>
__device__ __forceinline__ void conv_tile(const ConvArgs& a_in, const int tiles_x, const int tiles_y, const int cout_tiles,
                                          const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order: consecutive block ids land on different XCDs, so give every
    // XCD a contiguous run of tiles (cout tiles of one pixel tile share its input in L2).
    int lid;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // conv11 as four 2x2 parity convolutions in ONE launch (ConvArgs::n_par == 4): the four parities of a tile are
    // neighbouring workgroups of one XCD, so the low-resolution input tile they all stage comes out of that L2 once, and
    // a small batch gives the chip 4x the workgroups per launch.  Everything that depends on the parity is patched into
    // a private copy of the arguments (wave-uniform values); other kernels use the launch arguments as they are.
    ConvArgs a_par;
    if constexpr (C::KS == 2) {
        a_par = a_in;
        if (a_in.n_par == 4) {
            const int pl = lid & 3;
            lid >>= 2;
            a_par.sub_y = pl >> 1;
            a_par.sub_x = pl & 1;
            a_par.w = a_in.w_par[pl];
            a_par.amax_part0 = a_in.amax_part0 + pl * a_in.amax_par_stride;
        }
    }
    const ConvArgs& a = C::KS == 2 ? a_par : a_in;
    // split-K tiles may also split the input channels over `kparts` workgroups (grid = tiles x kparts, the parts of a
    // tile adjacent): each sums its channel range, the last one to finish adds the partial tiles in part order
    int kp = 0;
    if constexpr (C::SPLITK) {
        if (a.kparts > 1) {
            kp = lid % a.kparts;
            lid /= a.kparts;
        }
    }
    const int tile_id = lid;
    const int ct = lid % cout_tiles;
    const int pt = lid / cout_tiles;
    const int tx = pt % tiles_x;
    const int ty = (pt / tiles_x) % tiles_y;
    const int tb = pt / (tiles_x * tiles_y);
    const int x0 = tx * C::TW, y0 = ty * C::TRI, b0 = tb * C::NIMG, co0 = ct * C::COUT_T;

#if defined(MVLM_CONV_TIMING)
    const long long t_start = clock64();
#endif
    const int H = a.H, W = a.W;
    const int Hin = a.up_in ? (H >> 1) : H, Win = a.up_in ? (W >> 1) : W;
    const unsigned HWin = unsigned(Hin) * unsigned(Win);

    // ---- per-thread staging plan for the input tile (fixed across K-chunks) ----------
    unsigned goff[C::X_ITERS];
    unsigned goff2[C::X_ITERS];  // IN2: offsets into the half-resolution tensor
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
        goff2[i] = (IN2 && ok) ? (unsigned(b * a.in2_ctot + c) * (HWin >> 2) + unsigned((ys >> 1) * (Win >> 1) + (xs >> 1))) : 0u;
    }
    // weight slice: float4 index f -> (tap, c, cout4); the source offset is chunk-invariant
    // apart from the channel base
    unsigned woff_g[C::W_ITERS];
#pragma unroll
    for (int i = 0; i < C::W_ITERS; ++i) {
        const int f = tid + i * 256;
        const int row = f / (C::COUT_T / 4);
        const int c4 = f - row * (C::COUT_T / 4);
        const int tap = row / C::CK;
        const int c = row - tap * C::CK;
        woff_g[i] = (f < C::WT / 4) ? unsigned((tap * a.cin_pad + c) * a.cout_pad + co0 + c4 * 4) : INVALID_OFF;
    }

    // ---- per-lane operand offsets inside a stage ---------------------------------------
    int pixoff[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int p = (C::SPLITK ? 0 : wave * (C::PIX_T / 4)) + n * 32 + l31;
        const int x = p % C::TW;
        const int rr = p / C::TW;
        const int yl = rr % C::TRI;
        const int img = rr / C::TRI;
        pixoff[n] = (img * C::PH + yl) * C::PW + x + half * C::PLANE + (C::KS == 2 ? a.sub_y * C::PW + a.sub_x : 0) +
                    (C::SPLITK ? wave * C::CKW * C::PLANE : 0);
    }
    const int woff = C::XT_PAD + half * C::COUT_T + l31 + (C::SPLITK ? wave * C::CKW * C::COUT_T : 0);

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n) acc[m][n] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // 16-row strip (TAIL16): lane l multiplies channel 32*MT + (l & 15) with k = l >> 4 of a tap's four
    // channels; result register i of column group j is channel 32*MT + 4*(l >> 4) + i at pixel 16*j + (l & 15)
    const int q16 = lane >> 4, i16 = lane & 15;
    const int woff16 = C::XT_PAD + q16 * C::COUT_T + C::MT * 32 + i16;
    int pixoff16[C::NT16];
    f32x4 acc16[C::NT16];
#pragma unroll
    for (int j = 0; j < C::NT16; ++j) {
        const int p = wave * (C::PIX_T / 4) + j * 16 + i16;
        pixoff16[j] = (p / C::TW % C::TRI) * C::PW + p % C::TW + q16 * C::PLANE + (C::KS == 2 ? a.sub_y * C::PW + a.sub_x : 0);
        acc16[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    Strip4 s4;
    if constexpr (C::TAIL4) {
        const int p = wave * (C::PIX_T / 4) + lane;  // this lane's pixel of the tile
        s4.woff4 = C::XT_PAD + C::MT * 32 + 16 + (lane & 3);
        s4.pixoff4 = (p / C::TW % C::TRI) * C::PW + p % C::TW + (C::KS == 2 ? a.sub_y * C::PW + a.sub_x : 0);
    }
    // The consumer-side BatchNorm (scale, shift per input channel) is read from an LDS copy.  The first
    // chunk takes its parameters straight from global memory, so the table fill, the first input tile
    // and the first weight slice are ONE memory round trip, closed by the barrier below.
    float* sbn = smem + 2 * C::STAGE;
    StageRegs<C> regs;
    constexpr int T_TOT = C::X_ITERS + C::W_ITERS;
    int cb0 = 0, cb1 = a.cin_pad;  // this workgroup's input channels
    if constexpr (C::SPLITK) {
        const int span = a.cin_pad / a.kparts;
        cb0 = kp * span;
        cb1 = cb0 + span;
    }
    // Split-K tiles (one 32x32 tile per workgroup, no K parts over workgroups): after the reduction wave w owns the
    // accumulator registers 4w..4w+3 = the channels co0 + 8w + 4 half + (0..3) and finishes them alone, so the four
    // waves' epilogues run side by side.  Its residual values are requested HERE, before the K loop: the epilogue of
    // these latency-bound launches then adds and stores without a memory round trip of its own.
    SplitKTails<C::SPLITK ? C::NT : 1> sks;
    if constexpr (C::SPLITK) {
        if (a.kparts <= 1) {
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                SplitKTail& sk = sks.t[n];
                const int p = n * 32 + l31;
                const int rr = p / C::TW;
                const int b = b0 + rr / C::TRI;
                const int y = y0 + rr % C::TRI, x = x0 + p % C::TW;
                sk.ok = C::NIMG == 1 ? true : (b < a.B);
                sk.b = sk.ok ? unsigned(b) : 0u;
                sk.pix = unsigned(y * a.W + x);
                sk.y = y;
                sk.x = x;
                const unsigned HWo = unsigned(a.H) * unsigned(a.W);
                const int cl = co0 + 8 * wave + 4 * half;  // this lane's first channel
                if (a.res1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cs = (cl + j < a.cout) ? cl + j : 0;  // padded channel rows read inside the tensor
                        sk.res[j] = a.res1[(size_t(sk.b) * a.res1_ctot + a.res1_coff + cs) * HWo + sk.pix];
                    }
                    if (a.res2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int cs = (cl + j < a.cout) ? cl + j : 0;
                            sk.res[j] += a.res2[(size_t(sk.b) * a.res2_ctot + a.res2_coff + cs) * HWo + sk.pix];
                        }
                    }
                }
            }
        }
    }
    static_for<0, T_TOT>([&](auto tc) { issue_item<C, decltype(tc)::value, true, IN2>(a, cb0, tid, HWin, sbn, goff, woff_g, regs, smem, goff2); });
    if (a.pre_scale != nullptr) {
        for (int i = tid; i < a.cin_pad; i += 256) {
            sbn[i] = a.pre_scale[i];
            sbn[C::BN_MAXC + i] = a.pre_shift[i];
        }
    }
    static_for<0, T_TOT>([&](auto tc) { write_item<C, decltype(tc)::value, IN2>(a, cb0, tid, smem, goff, regs); });
    __syncthreads();

#if defined(MVLM_CONV_TIMING)
    const long long t_loop = clock64();
#endif
    int cur = 0;
    for (int cb = cb0 + C::CK; cb < cb1; cb += C::CK) {
#if defined(MVLM_ABLATE_NO_STAGING)  // timing experiment only: wrong results
        compute_chunk<C, false, AMAX, IN2>(a, smem + cur * C::STAGE, smem + (cur ^ 1) * C::STAGE, cb, tid, HWin, sbn, woff, pixoff,
                                goff, woff_g, regs, acc, woff16, pixoff16, acc16, s4, goff2);
#else
        compute_chunk<C, true, AMAX, IN2>(a, smem + cur * C::STAGE, smem + (cur ^ 1) * C::STAGE, cb, tid, HWin, sbn, woff, pixoff,
                               goff, woff_g, regs, acc, woff16, pixoff16, acc16, s4, goff2);
#endif
#if !defined(MVLM_ABLATE_NO_BARRIER)
        __syncthreads();  // next stage complete; everybody is done reading this one
#endif
        cur ^= 1;
    }
    compute_chunk<C, false, AMAX, IN2>(a, smem + cur * C::STAGE, nullptr, 0, tid, HWin, sbn, woff, pixoff, goff, woff_g, regs, acc, woff16, pixoff16, acc16, s4, goff2);
#if defined(MVLM_CONV_TIMING)
    const long long t_epi = clock64();
#endif

#if defined(MVLM_ABLATE_NO_EPILOGUE)  // timing experiment only: wrong results
    {
        float sacc = 0.f;
        static_for<0, C::MT>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
#pragma unroll
            for (int n = 0; n < C::NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[m][n][r];
        });
        if (sacc == 123.456f) a.out[0] = sacc;
        return;
    }
#endif
    if constexpr (C::SPLITK && !AMAX) {
        if (a.kparts <= 1) {
            // every wave leaves its partial tile in LDS, then wave w adds registers 4w..4w+3 of the four partials in wave
            // order ((w0 + w1) + w2) + w3 - the order of the single-wave reduction below, bit for bit - and finishes them;
            // a two-column tile does that column by column through the same buffer
            static_assert(!C::SPLITK || 2 * C::STAGE >= 4 * 16 * 64, "the four partial tiles must fit the two stages");
            float* const red = smem;
            const unsigned HWo = unsigned(a.H) * unsigned(a.W);
            const int cl = co0 + 8 * wave + 4 * half;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + cl);  // padded to cout_pad: never out of bounds
            if (a.post_scale) {
                ps4 = *reinterpret_cast<const f32x4*>(a.post_scale + cl);
                pt4 = *reinterpret_cast<const f32x4*>(a.post_shift + cl);
            }
            static_for<0, C::NT>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                const SplitKTail& sk = sks.t[n];
                __syncthreads();  // every wave is done reading the stages (or the previous column's partials): reuse them as the exchange buffer
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[0][n][r];
                __syncthreads();
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = red[(0 * 16 + 4 * wave + j) * 64 + lane];
                    t += red[(1 * 16 + 4 * wave + j) * 64 + lane];
                    t += red[(2 * 16 + 4 * wave + j) * 64 + lane];
                    t += red[(3 * 16 + 4 * wave + j) * 64 + lane];
                    v[j] = t;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (a.bias) v[j] += bias4[j];
                    if (a.post_scale) v[j] = fmaxf(fmaf(v[j], ps4[j], pt4[j]), 0.f);
                }
                if (a.out_raw) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (sk.ok && cl + j < a.cout) a.out_raw[(size_t(sk.b) * a.raw_ctot + a.raw_coff + cl + j) * HWo + sk.pix] = v[j];
                }
                if (a.res1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += sk.res[j];
                }
                if constexpr (C::POOL_SMALL) {
                    if (a.pool_out) {  // (wave-uniform: every lane takes part in the exchanges)
                        const unsigned o_pool = unsigned((sk.y >> 1) * (a.W >> 1) + (sk.x >> 1));
                        const bool writer = sk.ok && (l31 & C::TW) == 0 && (l31 & 1) == 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float m1 = fmaxf(v[j], __shfl_xor(v[j], C::TW));  // the row below / above
                            const int mi = __float_as_int(m1);
                            const float m2 = fmaxf(m1, __int_as_float(__builtin_amdgcn_update_dpp(mi, mi, 0xB1, 0xf, 0xf, false)));  // lane ^ 1
                            if (writer && cl + j < a.cout) a.pool_out[(size_t(sk.b) * a.pool_ctot + a.pool_coff + cl + j) * (HWo / 4) + o_pool] = m2;
                        }
                    }
                }
                if (a.out) {
                    if (!a.up_out) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (sk.ok && cl + j < a.cout) a.out[(size_t(sk.b) * a.out_ctot + a.out_coff + cl + j) * HWo + sk.pix] = v[j];
                    } else if (a.up_out == 2) {
                        const unsigned o2 = unsigned(2 * sk.y + a.sub_y) * unsigned(2 * a.W) + unsigned(2 * sk.x + a.sub_x);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (sk.ok && cl + j < a.cout) a.out[(size_t(sk.b) * a.out_ctot + a.out_coff + cl + j) * (4u * HWo) + o2] = v[j];
                    } else {
                        // hourglass up-path: the value goes to its 2x2 block of the skip tensor, added in place (:334-359)
                        const unsigned W2 = 2u * unsigned(a.W);
                        const unsigned o2 = unsigned(2 * sk.y) * W2 + unsigned(2 * sk.x);
                        float2 s0[4], s1[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int cs = (cl + j < a.cout) ? cl + j : 0;
                            const float* const ps = a.skip + (size_t(sk.b) * a.skip_ctot + a.skip_coff + cs) * (4u * HWo) + o2;
                            s0[j] = *reinterpret_cast<const float2*>(ps);
                            s1[j] = *reinterpret_cast<const float2*>(ps + W2);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (sk.ok && cl + j < a.cout) {
                                float* const po = a.out + (size_t(sk.b) * a.out_ctot + a.out_coff + cl + j) * (4u * HWo) + o2;
                                *reinterpret_cast<float2*>(po) = make_float2(v[j] + s0[j].x, v[j] + s0[j].y);
                                *reinterpret_cast<float2*>(po + W2) = make_float2(v[j] + s1[j].x, v[j] + s1[j].y);
                            }
                        }
                    }
                }
            });
            return;
        }
    }
    if constexpr (C::SPLITK) {
        // (input channels also divided over workgroups, or a fused-argmax launch) add the four waves' partial tiles in wave
        // order (deterministic), wave 0 finishes alone
        __syncthreads();  // every wave is done reading the stages: reuse them as the exchange buffer
        float* red = smem;
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(w * 16 + r) * 64 + lane];
        if (a.kparts > 1) {
            // partial tile -> workspace; the workgroup that takes the last ticket of this tile sums all parts in part
            // order (its own from memory too, so the order does not depend on who arrives last) and runs the epilogue
            float* const mine = a.kws + (size_t(tile_id) * a.kparts + kp) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = acc[0][0][r];
            __threadfence();  // release: the partial is visible device-wide before the ticket
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(a.kcnt + tile_id, 1u);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket != unsigned(a.kparts - 1)) return;
            __threadfence();  // acquire: the other parts' stores
            if (lane == 0) a.kcnt[tile_id] = 0;  // ready for the next launch on this stream
            const float* const all = a.kws + size_t(tile_id) * a.kparts * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = __builtin_nontemporal_load(all + r * 64 + lane);
            for (int q = 1; q < a.kparts; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += __builtin_nontemporal_load(all + q * 1024 + r * 64 + lane);
        }
    }

    if constexpr (AMAX) {
        // Fused heatmap argmax (paulsenpredictor.py:123) on the TRANSPOSED accumulators: conv11 has no residual /
        // post-BN, so the heatmap value is acc + bias.  Lane (half, l31) owns output channel co0 + 32 m + l31 and, of
        // every 32-pixel row segment n of its wave, the pixels x0 + (r & 3) + 8 (r >> 2) + 4 half: the maximum over a
        // lane's pixels is a chain of compares on registers (np.argmax order through argmax_key: NaN = maximum,
        // first maximum in row-major order wins), one exchange between the half-waves finishes a wave's partial.
        const int W = a.W;
        constexpr int ROWS_PER_WAVE = C::PIX_T / 4 / C::TW;  // = NT for the 32-pixel-row tiles
        static_assert(C::TW == 32 && C::NIMG == 1 && ROWS_PER_WAVE == C::NT, "fused argmax expects 32-pixel row tiles");
        const size_t part = size_t(a.amax_part0) + (size_t(ty) * tiles_x + tx) * 4 + wave;
        auto pixel_index = [&](int y, int x) {
            return a.up_out == 2 ? (2 * y + a.sub_y) * (2 * W) + 2 * x + a.sub_x : y * W + x;
        };
        static_for<0, C::MT>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const int co = co0 + m * 32 + l31;
            const float bias = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
            int best_k = int(0x807fffff);  // argmax_key(-inf)
            int best_i = 0x7fffffff;
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const int y = y0 + wave * ROWS_PER_WAVE + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int k = argmax_key(acc[m][n][r] + bias);
                    const int pix = pixel_index(y, x);
                    const bool better = k > best_k;  // ascending pixel order within the lane: strict > keeps the first
                    best_k = better ? k : best_k;
                    best_i = better ? pix : best_i;
                }
            }
            // the other half-wave holds the other pixels of the same channel
            const int ok = __shfl_xor(best_k, 32), oi = __shfl_xor(best_i, 32);
            if (ok > best_k || (ok == best_k && oi < best_i)) {
                best_k = ok;
                best_i = oi;
            }
            if (half == 0 && co < a.cout && b0 < a.B) {
                const size_t o = (size_t(b0) * a.cout + co) * a.amax_parts + part;
                a.amax_val[o] = argmax_value(best_k);
                a.amax_idx[o] = best_i;
            }
        });
        if constexpr (C::TAIL16) {
            // strip (transposed 16x16x4 tiles): lane (q16, i16) owns channel co0 + 32 MT + i16 and the pixels
            // 16 j + 4 q16 + i of its wave's pixel range
            const int co = co0 + C::MT * 32 + i16;
            const float bias = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
            int best_k = int(0x807fffff);
            int best_i = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < C::NT16; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = wave * (C::PIX_T / 4) + j * 16 + 4 * q16 + i;
                    const int k = argmax_key(acc16[j][i] + bias);
                    const int pix = pixel_index(y0 + p / C::TW, x0 + p % C::TW);
                    const bool better = k > best_k || (k == best_k && pix < best_i);
                    best_k = better ? k : best_k;
                    best_i = better ? pix : best_i;
                }
#pragma unroll
            for (int s = 16; s <= 32; s <<= 1) {
                const int ok = __shfl_xor(best_k, s), oi = __shfl_xor(best_i, s);
                if (ok > best_k || (ok == best_k && oi < best_i)) {
                    best_k = ok;
                    best_i = oi;
                }
            }
            if (q16 == 0 && co < a.cout && b0 < a.B) {
                const size_t o = (size_t(b0) * a.cout + co) * a.amax_parts + part;
                a.amax_val[o] = argmax_value(best_k);
                a.amax_idx[o] = best_i;
            }
        }
        if constexpr (C::TAIL4) {
            // 4-row strip: register v = channel co0 + 32 MT + 16 + v at THIS lane's pixel (64 consecutive pixels of the
            // tile per wave): the wave's partial of a channel is a 64-lane reduction, first maximum in row-major order
            const int p = wave * (C::PIX_T / 4) + lane;
            const int pix = pixel_index(y0 + p / C::TW, x0 + p % C::TW);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int co = co0 + C::MT * 32 + 16 + v;
                const float bias = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
                int best_k = argmax_key(s4.acc[v] + bias), best_i = pix;
#pragma unroll
                for (int sft = 1; sft <= 32; sft <<= 1) {
                    const int ok = __shfl_xor(best_k, sft), oi = __shfl_xor(best_i, sft);
                    if (ok > best_k || (ok == best_k && oi < best_i)) {
                        best_k = ok;
                        best_i = oi;
                    }
                }
                if (lane == 0 && co < a.cout && b0 < a.B) {
                    const size_t o = (size_t(b0) * a.cout + co) * a.amax_parts + part;
                    a.amax_val[o] = argmax_value(best_k);
                    a.amax_idx[o] = best_i;
                }
            }
        }
        return;
    }

    // ---------------------------------- epilogue ---------------------------------------
    // Addresses are wave-uniform channel bases (scalar registers) + one 32-bit per-lane offset per
    // tensor and pixel column, so an element costs a load/add/store, not 64-bit vector arithmetic.
    const unsigned HW = unsigned(H) * unsigned(W);
    bool lane_ok[C::NT];
    int ppix[C::NT];
    unsigned o_raw[C::NT], o_r1[C::NT], o_r2[C::NT], o_out[C::NT], o_skip[C::NT];
    // fused 2x2 max-pool: the rows of a pair are two pixel registers of one lane (n, n+1), the columns two
    // neighbouring lanes, so it needs 32-pixel tile rows, one image per tile and an even row count per wave
    constexpr bool CAN_POOL = !C::SPLITK && C::TW == 32 && C::NIMG == 1 && C::NT % 2 == 0;
    constexpr bool POOLS = C::POOL_SMALL;  // narrow tiles: partner lanes l ^ TW (row) and l ^ 1 (column), see Cfg
    unsigned o_pool[C::NT / 2 > 0 ? C::NT / 2 : 1];
    unsigned o_pool_s[POOLS ? C::NT : 1];
    const bool pool_writer = (l31 & (C::TW & 31)) == 0 && (l31 & 1) == 0;
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int p = (C::SPLITK ? 0 : wave * (C::PIX_T / 4)) + n * 32 + l31;
        const int rr = p / C::TW;
        const int b = b0 + rr / C::TRI;
        const int y = y0 + rr % C::TRI, x = x0 + p % C::TW;
        lane_ok[n] = C::NIMG == 1 ? true : (b < a.B);
        const unsigned bb = lane_ok[n] ? unsigned(b) : 0u;
        const unsigned pix = unsigned(y * W + x);
        const unsigned h4 = 4u * unsigned(half);
        ppix[n] = int(pix);
        o_raw[n] = (bb * a.raw_ctot + a.raw_coff + h4) * HW + pix;
        o_r1[n] = (bb * a.res1_ctot + a.res1_coff + h4) * HW + pix;
        o_r2[n] = (bb * a.res2_ctot + a.res2_coff + h4) * HW + pix;
        if constexpr (CAN_POOL)
            if ((n & 1) == 0) o_pool[n / 2] = (bb * a.pool_ctot + a.pool_coff + h4) * (HW / 4) + unsigned((y >> 1) * (W >> 1) + (x >> 1));
        if constexpr (POOLS) o_pool_s[n] = (bb * a.pool_ctot + a.pool_coff + h4) * (HW / 4) + unsigned((y >> 1) * (W >> 1) + (x >> 1));
        if (!a.up_out) {
            o_out[n] = (bb * a.out_ctot + a.out_coff + h4) * HW + pix;
            o_skip[n] = 0;
        } else if (a.up_out == 2) {
            const unsigned o2 = unsigned(2 * y + a.sub_y) * unsigned(2 * W) + unsigned(2 * x + a.sub_x);
            o_out[n] = (bb * a.out_ctot + a.out_coff + h4) * (4u * HW) + o2;
            o_skip[n] = 0;
            ppix[n] = int(o2);  // the argmax runs over full-resolution pixel indices
        } else {
            const unsigned o2 = unsigned(2 * y) * unsigned(2 * W) + unsigned(2 * x);
            o_out[n] = (bb * a.out_ctot + a.out_coff + h4) * (4u * HW) + o2;
            o_skip[n] = (bb * a.skip_ctot + a.skip_coff + h4) * (4u * HW) + o2;
        }
    }
    // Channels are walked in groups of four accumulator registers (= four consecutive channels
    // per half-wave).  The residual values of group g+1 are loaded (unconditionally, clamped
    // addresses) before group g is finished and stored, so a load's latency is paid once per
    // group of 4*NT elements instead of once per element, and the feature flags cost one
    // wave-uniform branch per group.
    constexpr int GE = 4 * C::NT;       // elements per group
    constexpr int NG = C::MT * 4;       // groups per wave
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;  // every channel row of this tile exists
        // The tensors may share storage (a block adds into its own residual slice in place, the
        // hourglass adds into its skip tensor), but only element-wise: a value is stored to the
        // address it was loaded from, never to another group's.  __restrict__ tells the compiler
        // exactly that, so it stops draining the memory pipeline (s_waitcnt vmcnt(0)) between one
        // group's stores and the next group's loads.
        const float* __restrict__ const t_res1 = a.res1;
        const float* __restrict__ const t_res2 = a.res2;
        const float* __restrict__ const t_skip = a.skip;
        float* __restrict__ const t_raw = a.out_raw;
        float* __restrict__ const t_out = a.out;
        float* __restrict__ const t_pool = a.pool_out;
        float resv[2][GE];
        auto group_base = [&](int g) { return co0 + (g >> 2) * 32 + 8 * (g & 3); };  // wave-uniform first channel
        auto load_group = [&](auto gc, float (&dst)[GE]) {
            constexpr int g = decltype(gc)::value;
            if (a.res1) {
                const int cs0 = group_base(g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // clamp the row so padded channel rows still read inside the tensor
                    const int cs = (FULL || cs0 + j + 4 < a.cout) ? cs0 + j : 0;
                    const float* const p1 = t_res1 + size_t(cs) * HW;
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) dst[j * C::NT + n] = p1[o_r1[n]];
                }
                if (a.res2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cs = (FULL || cs0 + j + 4 < a.cout) ? cs0 + j : 0;
                        const float* const p2 = t_res2 + size_t(cs) * HW;
#pragma unroll
                        for (int n = 0; n < C::NT; ++n) dst[j * C::NT + n] += p2[o_r2[n]];
                    }
                }
            }
        };
        load_group(std::integral_constant<int, 0>{}, resv[0]);
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int m = g >> 2, rg = g & 3;
            if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, resv[(g + 1) & 1]);
            // keep the next group's loads HERE: the scheduler otherwise sinks them to their first use and
            // every group pays a full memory round trip
            __builtin_amdgcn_sched_barrier(0);
            const int cs0 = group_base(g);
            // bias / post-BN parameters of this lane's four channels: ONE 16-byte load each (the
            // vectors are padded to cout_pad, so rows beyond cout read zeros, never out of bounds)
            const int cl = cs0 + 4 * half;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + cl);
            if (a.post_scale) {
                ps4 = *reinterpret_cast<const f32x4*>(a.post_scale + cl);
                pt4 = *reinterpret_cast<const f32x4*>(a.post_shift + cl);
            }
            float vals[GE];
            bool okc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                okc[j] = FULL || cl + j < a.cout;
#pragma unroll
                for (int n = 0; n < C::NT; ++n) {
                    float v = acc[m][n][4 * rg + j];
                    if (a.bias) v += bias4[j];
                    if (a.post_scale) v = fmaxf(fmaf(v, ps4[j], pt4[j]), 0.f);
                    vals[j * C::NT + n] = v;
                }
            }
            if (a.out_raw) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* const p = t_raw + size_t(cs0 + j) * HW;
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        if (okc[j] && lane_ok[n]) p[o_raw[n]] = vals[j * C::NT + n];
                }
            }
            if (a.res1) {
#pragma unroll
                for (int e = 0; e < GE; ++e) vals[e] += resv[g & 1][e];
            }
            if constexpr (CAN_POOL) {
                if (a.pool_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* const p = t_pool + size_t(cs0 + j) * (HW / 4);
#pragma unroll
                        for (int q = 0; q < C::NT / 2; ++q) {
                            const float v2 = fmaxf(vals[j * C::NT + 2 * q], vals[j * C::NT + 2 * q + 1]);
                            const int vi = __float_as_int(v2);
                            const float other = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xf, 0xf, false));  // lane ^ 1
                            if (okc[j] && (l31 & 1) == 0) p[o_pool[q]] = fmaxf(v2, other);
                        }
                    }
                }
            }
            if constexpr (POOLS) {
                if (a.pool_out) {  // (wave-uniform)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* const p = t_pool + size_t(cs0 + j) * (HW / 4);
#pragma unroll
                        for (int n = 0; n < C::NT; ++n) {
                            const float m1 = fmaxf(vals[j * C::NT + n], __shfl_xor(vals[j * C::NT + n], C::TW));
                            const int mi = __float_as_int(m1);
                            const float m2 = fmaxf(m1, __int_as_float(__builtin_amdgcn_update_dpp(mi, mi, 0xB1, 0xf, 0xf, false)));  // lane ^ 1
                            if (okc[j] && lane_ok[n] && pool_writer) p[o_pool_s[n]] = m2;
                        }
                    }
                }
            }
            if (a.out) {
                if (a.up_out != 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* const p = t_out + size_t(cs0 + j) * HW * (a.up_out ? 4 : 1);
#pragma unroll
                        for (int n = 0; n < C::NT; ++n)
                            if (okc[j] && lane_ok[n]) p[o_out[n]] = vals[j * C::NT + n];
                    }
                } else {
                    const unsigned W2 = 2u * unsigned(W);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // the row's skip loads first, then its 2x2 scatter
                        const int cs = okc[j] ? cs0 + j : 0;
                        const float* const ps = t_skip + size_t(cs) * HW * 4;
                        float* const p = t_out + size_t(cs0 + j) * HW * 4;
                        float2 s0[C::NT], s1[C::NT];
#pragma unroll
                        for (int n = 0; n < C::NT; ++n) {
                            s0[n] = *reinterpret_cast<const float2*>(ps + o_skip[n]);
                            s1[n] = *reinterpret_cast<const float2*>(ps + o_skip[n] + W2);
                        }
#pragma unroll
                        for (int n = 0; n < C::NT; ++n) {
                            if (okc[j] && lane_ok[n]) {
                                const float v = vals[j * C::NT + n];
                                *reinterpret_cast<float2*>(p + o_out[n]) = make_float2(v + s0[n].x, v + s0[n].y);
                                *reinterpret_cast<float2*>(p + o_out[n] + W2) = make_float2(v + s1[n].x, v + s1[n].y);
                            }
                        }
                    }
                }
            }
        });
    };
    // Branch-free epilogues for the layer kinds that carry the time (full channel tiles, plain
    // NCHW output).  The general epilogue above tests its feature flags per group; at every join
    // of such a branch the compiler has to assume the worst about outstanding memory operations
    // and drains the pipeline (s_waitcnt vmcnt(0)) - which serialises "load residual, add, store"
    // into one memory round trip per group (measured: 74k of a c128 tile's 1.3M cycles).  With
    // the flags as template parameters every load and store is unconditional, the waits are exact
    // and the residual / parameter prefetch really runs ahead.
    //   RAW: also store the pre-residual value (input of the block's next conv)
    //   RES: add the residual slice        PAR: 0 none, 1 bias, 2 bias + post-BatchNorm + ReLU
    //   POOL: also store the 2x2 max-pooled value
    //   RES 2: two residuals, (res1 + res2) + value (conv7: r3 + ll1 + conv7(x), paulsenpredictor.py:422)
    //   SCAT: the hourglass up-path - the value goes to its 2x2 block of the skip tensor, added in place (:334-359)
    auto epilogue_fast = [&](auto raw_c, auto res_c, auto par_c, auto pool_c, auto scat_c) __attribute__((always_inline)) {
        constexpr bool RAW = decltype(raw_c)::value, RES = decltype(res_c)::value != 0, RES2 = decltype(res_c)::value == 2;
        constexpr bool POOL = decltype(pool_c)::value != 0, FULL_OUT = decltype(pool_c)::value != 2;  // 2: pooled tensor only
        constexpr bool SCAT = decltype(scat_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        // residual prefetch distance in groups (measured: 3 and 5 perform alike, 6 spills on the
        // 128-accumulator tiles)
        constexpr int PF_WANT = (GE <= 8 && !RES2 && !SCAT) ? 3 : 1;  // (two residual rings / the skip blocks: registers)
        constexpr int PF = RES ? (PF_WANT < NG ? PF_WANT : NG) : 0;
        constexpr int RING = PF + 1;
        float resv[RING][GE];
        float resv2[RES2 ? RING : 1][GE];
        f32x4 bias_r[2], ps_r[2], pt_r[2];
        auto chan0 = [&](int g) { return co0 + (g >> 2) * 32 + 8 * (g & 3); };  // wave-uniform first channel
        auto ld_res = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (RES) {
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const float* const p1 = a.res1 + size_t(chan0(g) + j) * HW;
                    static_for<0, C::NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        resv[g % RING][j * C::NT + n] = p1[o_r1[n]];
                    });
                    if constexpr (RES2) {
                        const float* const p2 = a.res2 + size_t(chan0(g) + j) * HW;
                        static_for<0, C::NT>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            resv2[g % RING][j * C::NT + n] = p2[o_r2[n]];
                        });
                    }
                });
            }
        };
        auto ld_par = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (PAR >= 1) bias_r[g & 1] = *reinterpret_cast<const f32x4*>(a.bias + chan0(g) + 4 * half);
            if constexpr (PAR == 2) {
                ps_r[g & 1] = *reinterpret_cast<const f32x4*>(a.post_scale + chan0(g) + 4 * half);
                pt_r[g & 1] = *reinterpret_cast<const f32x4*>(a.post_shift + chan0(g) + 4 * half);
            }
        };
        static_for<0, (PF < NG ? PF : NG)>([&](auto gc) { ld_res(gc); });
        ld_par(std::integral_constant<int, 0>{});
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int m = g >> 2, rg = g & 3;
            if constexpr (g + PF < NG) ld_res(std::integral_constant<int, g + PF>{});
            if constexpr (g + 1 < NG) ld_par(std::integral_constant<int, g + 1>{});
            const int cs0 = chan0(g);
            // the group's 2x2 blocks of the skip tensor, SK_CH channels at a time, requested before those channels' other
            // work (two where the registers allow it: pipelining a channel ahead, or a whole group at once, spilled on the
            // 168-register tiles)
            constexpr int SK_CH = (C::MIN_BLOCKS_PER_CU == 2 && C::NT <= 2) ? 2 : 1;
            float2 sk0[SCAT ? SK_CH * C::NT : 1], sk1[SCAT ? SK_CH * C::NT : 1];
            auto ld_skip = [&](auto hc) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;  // channels SK_CH h ... of the group
                if constexpr (SCAT) {
                    const unsigned W2 = 2u * unsigned(W);
                    static_for<0, SK_CH>([&](auto jc) {
                        constexpr int jj = decltype(jc)::value, j = SK_CH * h + jj;
                        const float* const ps = a.skip + size_t(cs0 + j) * HW * 4;
                        static_for<0, C::NT>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            sk0[jj * C::NT + n] = *reinterpret_cast<const float2*>(ps + o_skip[n]);
                            sk1[jj * C::NT + n] = *reinterpret_cast<const float2*>(ps + o_skip[n] + W2);
                        });
                    });
                }
            };
            if constexpr (SK_CH == 2) ld_skip(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);  // the prefetches stay ahead of this group's work
            float vals[GE];
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<0, C::NT>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    float v = acc[m][n][4 * rg + j];
                    if constexpr (PAR >= 1) v += bias_r[g & 1][j];
                    if constexpr (PAR == 2) v = fmaxf(fmaf(v, ps_r[g & 1][j], pt_r[g & 1][j]), 0.f);
                    vals[j * C::NT + n] = v;
                });
            });
            if constexpr (RAW) {
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float* const p = a.out_raw + size_t(cs0 + j) * HW;
                    static_for<0, C::NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        if (lane_ok[n]) p[o_raw[n]] = vals[j * C::NT + n];
                    });
                });
            }
            if constexpr (RES2) {
                static_for<0, GE>([&](auto ec) { vals[decltype(ec)::value] += resv[g % RING][decltype(ec)::value] + resv2[g % RING][decltype(ec)::value]; });
            } else if constexpr (RES) {
                static_for<0, GE>([&](auto ec) { vals[decltype(ec)::value] += resv[g % RING][decltype(ec)::value]; });
            }
            if constexpr (POOL && CAN_POOL) {
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float* const p = a.pool_out + size_t(cs0 + j) * (HW / 4);
                    static_for<0, C::NT / 2>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        const float v2 = fmaxf(vals[j * C::NT + 2 * q], vals[j * C::NT + 2 * q + 1]);
                        const int vi = __float_as_int(v2);
                        const float other = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xf, 0xf, false));  // lane ^ 1
                        if ((l31 & 1) == 0) p[o_pool[q]] = fmaxf(v2, other);
                    });
                });
            }
            if constexpr (POOL && POOLS) {
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float* const p = a.pool_out + size_t(cs0 + j) * (HW / 4);
                    static_for<0, C::NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        const float m1 = fmaxf(vals[j * C::NT + n], __shfl_xor(vals[j * C::NT + n], C::TW));
                        const int mi = __float_as_int(m1);
                        const float m2 = fmaxf(m1, __int_as_float(__builtin_amdgcn_update_dpp(mi, mi, 0xB1, 0xf, 0xf, false)));  // lane ^ 1
                        if (lane_ok[n] && pool_writer) p[o_pool_s[n]] = m2;
                    });
                });
            }
            if constexpr (SCAT) {
                const unsigned W2 = 2u * unsigned(W);
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (SK_CH == 1) ld_skip(jc);
                    if constexpr (SK_CH == 2 && j == 2) ld_skip(std::integral_constant<int, 1>{});
                    float* const p = a.out + size_t(cs0 + j) * HW * 4;
                    static_for<0, C::NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        if (lane_ok[n]) {
                            const float v = vals[j * C::NT + n];
                            const float2 s0 = sk0[(j % SK_CH) * C::NT + n], s1 = sk1[(j % SK_CH) * C::NT + n];
                            *reinterpret_cast<float2*>(p + o_out[n]) = make_float2(v + s0.x, v + s0.y);
                            *reinterpret_cast<float2*>(p + o_out[n] + W2) = make_float2(v + s1.x, v + s1.y);
                        }
                    });
                });
            } else if constexpr (FULL_OUT) {
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float* const p = a.out + size_t(cs0 + j) * HW;
                    static_for<0, C::NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        if (lane_ok[n]) p[o_out[n]] = vals[j * C::NT + n];
                    });
                });
            }
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P2 = std::integral_constant<int, 2>;
    const bool full_tile = co0 + C::COUT_T <= a.cout;
    const bool has_raw = a.out_raw != nullptr, has_res = a.res1 != nullptr, has_pool = a.pool_out != nullptr;
    const bool scat = a.up_out == 1 && a.skip != nullptr && a.out != nullptr;
    const bool fast_ok = full_tile && !C::SPLITK && (a.out || has_pool) && (!a.up_out || scat) && (!a.skip || scat) && (CAN_POOL || POOLS || !has_pool);
    const int par = a.post_scale ? (a.bias ? 2 : -1) : (a.bias ? 1 : 0);
    const int pool_mode = has_pool ? (a.out ? 1 : 2) : 0;
    const int res_mode = has_res ? (a.res2 ? 2 : 1) : 0;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using P1 = std::integral_constant<int, 1>;
    const bool plain = fast_ok && !scat;  // the layer kinds of rounds 1-3
    if (plain && has_raw && res_mode == 1 && par == 0 && pool_mode == 0)
        epilogue_fast(T_{}, R1{}, P0{}, M0{}, F_{});  // block conv1 / conv2
    else if (plain && !has_raw && res_mode == 1 && par == 0 && pool_mode == 0)
        epilogue_fast(F_{}, R1{}, P0{}, M0{}, F_{});  // block conv3
    else if (plain && has_raw && res_mode == 1 && par == 0 && pool_mode == 1)
        epilogue_fast(T_{}, R1{}, P0{}, M1{}, F_{});  // ... of a block that is pooled next
    else if (plain && !has_raw && res_mode == 1 && par == 0 && pool_mode == 1)
        epilogue_fast(F_{}, R1{}, P0{}, M1{}, F_{});
    else if (plain && has_raw && res_mode == 1 && par == 0 && pool_mode == 2)
        epilogue_fast(T_{}, R1{}, P0{}, M2{}, F_{});  // ... whose full-resolution output nobody reads (stem)
    else if (plain && !has_raw && res_mode == 1 && par == 0 && pool_mode == 2)
        epilogue_fast(F_{}, R1{}, P0{}, M2{}, F_{});
    else if (plain && !has_raw && res_mode == 0 && par == 2 && pool_mode == 0)
        epilogue_fast(F_{}, R0{}, P2{}, M0{}, F_{});  // conv1, conv5, conv9
    else if (plain && !has_raw && res_mode == 0 && par == 0 && pool_mode == 0)
        epilogue_fast(F_{}, R0{}, P0{}, M0{}, F_{});  // 1x1 resample
    // (the two kinds below only where such a layer can run - 3x3 tiles without a strip: every kind compiled into a tile
    //  costs it registers, and the 80-row tile fell from three resident workgroups per CU to two with them: 166 -> 191)
    else if (C::KS == 3 && !C::TAIL16 && plain && !has_raw && res_mode == 2 && par == 1 && pool_mode == 1) {
        if constexpr (C::KS == 3 && !C::TAIL16) epilogue_fast(F_{}, R2{}, P1{}, M1{}, F_{});  // conv7 (+ the pooled copy the second hourglass starts from)
    } else if (C::KS == 3 && !C::TAIL16 && fast_ok && scat && has_raw && res_mode == 1 && par == 0 && pool_mode == 0) {
        if constexpr (C::KS == 3 && !C::TAIL16) epilogue_fast(T_{}, R1{}, P0{}, M0{}, T_{});  // conv1 / conv2 of a level's last block on the way up
    } else if (C::KS == 3 && !C::TAIL16 && fast_ok && scat && !has_raw && res_mode == 1 && par == 0 && pool_mode == 0) {
        if constexpr (C::KS == 3 && !C::TAIL16) epilogue_fast(F_{}, R1{}, P0{}, M0{}, T_{});  // ... its conv3
    }
    else if (full_tile)
        epilogue(std::true_type{});
    else
        epilogue(std::false_type{});
#if defined(MVLM_CONV_TIMING)
    if (a.timing && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);  // the epilogue's stores have left the wave
        const long long t_end = clock64();
        atomicAdd(a.timing + 0, (unsigned long long)(t_loop - t_start));
        atomicAdd(a.timing + 1, (unsigned long long)(t_epi - t_loop));
        atomicAdd(a.timing + 2, (unsigned long long)(t_end - t_epi));
        atomicAdd(a.timing + 3, 1ull);
    }
#endif

    // ---- the 16-row strip: plain conv + bias layers only (conv6, conv10, conv11; checked on the host) ----
    int pix16[C::NT16];            // output pixel index of column group j (full resolution for up_out == 2)
    float val16[C::NT16][4];       // acc + bias, kept for the fused argmax
    bool ok16[4];
    if constexpr (C::TAIL16) {
        const unsigned b = unsigned(b0);
#pragma unroll
        for (int j = 0; j < C::NT16; ++j) {
            const int p = wave * (C::PIX_T / 4) + j * 16 + i16;
            const int y = y0 + p / C::TW, x = x0 + p % C::TW;
            pix16[j] = a.up_out == 2 ? (2 * y + a.sub_y) * (2 * W) + 2 * x + a.sub_x : y * W + x;
        }
        const unsigned plane = a.up_out == 2 ? 4u * HW : HW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + C::MT * 32 + 4 * q16 + i;
            ok16[i] = co < a.cout;
            const int coc = ok16[i] ? co : 0;
            const float bias = a.bias ? a.bias[coc] : 0.f;
            float* const po = a.out ? a.out + (size_t(b) * a.out_ctot + a.out_coff + coc) * plane : nullptr;
#pragma unroll
            for (int j = 0; j < C::NT16; ++j) {
                val16[j][i] = acc16[j][i] + bias;
                if (po && ok16[i]) po[pix16[j]] = val16[j][i];
            }
        }
    }
    // ---- the 4-row strip: register v = channel co0 + 32 MT + 16 + v at this lane's pixel (64 consecutive pixels per wave)
    if constexpr (C::TAIL4) {
        const int p = wave * (C::PIX_T / 4) + lane;
        const int y = y0 + p / C::TW, x = x0 + p % C::TW;
        const unsigned pix = a.up_out == 2 ? unsigned(2 * y + a.sub_y) * unsigned(2 * W) + unsigned(2 * x + a.sub_x)
                                           : unsigned(y) * unsigned(W) + unsigned(x);
        const unsigned plane = a.up_out == 2 ? 4u * HW : HW;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int co = co0 + C::MT * 32 + 16 + v;
            if (co < a.cout && a.out)
                a.out[(size_t(b0) * a.out_ctot + a.out_coff + co) * plane + pix] = s4.acc[v] + (a.bias ? a.bias[co] : 0.f);
        }
    }

}

template <class C, bool AMAX, bool IN2 = false>
__global__ __launch_bounds__(256, C::MIN_BLOCKS_PER_CU) void conv_mfma_kernel(const ConvArgs a_in, const int tiles_x, const int tiles_y,
                                                           const int cout_tiles) {
    conv_tile<C, AMAX, IN2>(a_in, tiles_x, tiles_y, cout_tiles, int(blockIdx.x), int(gridDim.x));
}

// Two independent convolutions in ONE grid (same tile configuration; e.g. conv j of a hourglass level's skip block and
// conv j of the block that starts the next lower level - paulsenpredictor.py:301-361, up1 = rb(x) beside low1 =
// rb(pool(x))): workgroups [0, nblk[0]) belong to problem 0, [nblk0_pad, gridDim.x) to problem 1 (nblk0_pad = nblk[0] rounded
// up to a multiple of 8, so that blockIdx & 7 is the XCD in both ranges; the workgroups between exit).  A latency-bound
// launch of a small level then rides in the tail of the larger one without any cross-queue dependency, and the pair pays
// one grid fill and drain instead of two.  Results are those of the two separate launches with this variant, bit for bit.
struct ConvPairArgs {
    ConvArgs a[2];
    int tiles_x[2], tiles_y[2], cout_tiles[2], nblk[2];
    int nblk0_pad;
};

template <class C>
__global__ __launch_bounds__(256, C::MIN_BLOCKS_PER_CU) void conv_pair_kernel(const ConvPairArgs p) {
    const int bid = int(blockIdx.x);
    const int which = bid >= p.nblk0_pad ? 1 : 0;
    if (!which && bid >= p.nblk[0]) return;
    conv_tile<C, false>(p.a[which], p.tiles_x[which], p.tiles_y[which], p.cout_tiles[which], which ? bid - p.nblk0_pad : bid,
                        p.nblk[which]);
}

// ---- launchers -------------------------------------------------------------------------
struct ConvGrid {
    int tiles_x = 0, tiles_y = 0, cout_tiles = 0;
    long tiles = 0;  // output tiles (x parities); the grid has tiles * kparts workgroups
    long nblk = 0;
};

// host-side checks of one problem against a variant's tile geometry; fills the grid
template <class C>
int check_variant(mvlm_ctx* ctx, ConvArgs& a, ConvGrid& g) {
    const int tiles_x = a.W / C::TW, tiles_y = a.H / C::TRI;
    const int tiles_b = (a.B + C::NIMG - 1) / C::NIMG;
    const int cout_tiles = a.cout_pad / C::COUT_T;
    MVLM_REQUIRE(ctx, a.ksize == C::KS, "conv: kernel variant built for another kernel size");
    MVLM_REQUIRE(ctx, !a.in2 || (C::HAS_IN2 && !a.up_in && !a.amax_val && a.in_coff == 0 && a.in2_ctot >= a.cin && !(a.H & 1) && !(a.W & 1) && a.kparts <= 1),
                 "conv: a second input tensor (upsample + skip on the load) is served by the 128-channel 8x32 tile only");
    MVLM_REQUIRE(ctx, a.W % C::TW == 0 && a.H % C::TRI == 0, "conv: spatial size not a multiple of the tile");
    MVLM_REQUIRE(ctx, !C::SPLITK || a.cin_pad % 32 == 0, "conv: split-K tiles need 32-channel chunks");
    MVLM_REQUIRE(ctx, !(C::SPLITK && C::NT > 1) || (a.kparts <= 1 && !a.amax_val), "conv: the two-column split-K tiles have no K-parts / argmax form");
    MVLM_REQUIRE(ctx, a.cout_pad % C::COUT_T == 0, "conv: cout_pad not a multiple of the cout tile");
    MVLM_REQUIRE(ctx, a.cin_pad % C::CK == 0, "conv: cin_pad must be a multiple of the K-chunk");
    MVLM_REQUIRE(ctx, !a.pre_scale || a.cin_pad <= C::BN_MAXC, "conv: pre-activation BatchNorm supports up to 256 input channels");
    if (C::TAIL4) MVLM_REQUIRE(ctx, a.up_out != 1 && (a.out || (a.amax_val && C::HAS_AMAX)), "conv: the 84-channel tile writes a plain (or parity) output tensor or argmax partials");
    if (C::TAIL16)
        MVLM_REQUIRE(ctx, !a.res1 && !a.res2 && !a.out_raw && !a.post_scale && a.up_out != 1 && !a.pool_out,
                     "conv: the 80-channel tiles serve plain conv + bias layers only");
    if (a.amax_val) {
        MVLM_REQUIRE(ctx, C::NIMG == 1, "conv: fused argmax needs one image per tile");
        MVLM_REQUIRE(ctx, a.amax_part0 >= 0 && a.amax_part0 + tiles_x * tiles_y * 4 <= a.amax_parts,
                     "conv: argmax partial range mismatch");
        MVLM_REQUIRE(ctx, !a.res1 && !a.post_scale && a.up_out != 1, "conv: fused argmax expects a plain conv + bias layer");
        MVLM_REQUIRE(ctx, !a.out && !a.out_raw && !a.pool_out, "conv: a fused-argmax launch does not materialise the heatmap");
    }
    long nblk = long(tiles_x) * tiles_y * tiles_b * cout_tiles;
    if (a.n_par != 1) {
        MVLM_REQUIRE(ctx, a.n_par == 4 && C::KS == 2 && a.up_out == 2, "conv: four parities per launch are a 2x2-kernel feature");
        MVLM_REQUIRE(ctx, a.w_par[0] && a.w_par[1] && a.w_par[2] && a.w_par[3], "conv: parity weights missing");
        MVLM_REQUIRE(ctx, !a.amax_val || (a.amax_par_stride >= tiles_x * tiles_y * 4 &&
                                          a.amax_part0 + 3 * a.amax_par_stride + tiles_x * tiles_y * 4 <= a.amax_parts),
                     "conv: argmax partial range mismatch (four parities)");
        nblk *= 4;
    }
    MVLM_REQUIRE(ctx, nblk > 0 && nblk < (1l << 31), "conv: bad grid");
    a.kparts = a.kparts > 1 ? a.kparts : 1;
    g.tiles = nblk;
    if (a.kparts > 1) {
        MVLM_REQUIRE(ctx, C::SPLITK, "conv: only the split-K tiles divide the input channels over workgroups");
        MVLM_REQUIRE(ctx, a.cin_pad % (a.kparts * C::CK) == 0, "conv: input channels do not divide into the requested K parts");
        nblk *= a.kparts;
    }
    MVLM_REQUIRE(ctx, nblk < (1l << 31), "conv: bad grid");
    g.tiles_x = tiles_x;
    g.tiles_y = tiles_y;
    g.cout_tiles = cout_tiles;
    g.nblk = nblk;
    return 0;
}

// dynamic-LDS limit of this variant's kernels: set once per context, i.e. per device (hipFuncSetAttribute
// applies to the current device's copy of the function; the ctx mutex held by every entry point guards the mask)
template <class C>
int set_variant_attributes(mvlm_ctx* ctx, int variant_id) {
    if ((ctx->conv_attr_mask >> variant_id) & 1ull) return 0;
    MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<C, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES)));
    if constexpr (C::HAS_AMAX)
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<C, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES)));
    if constexpr (C::PAIRABLE)
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_kernel<C>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES)));
    if constexpr (C::HAS_IN2)
        MVLM_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<C, false, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES)));
    ctx->conv_attr_mask |= 1ull << variant_id;
    return 0;
}

template <class C>
int launch_variant(mvlm_ctx* ctx, const ConvArgs& a_in, int variant_id) {
    ConvArgs a = a_in;
#if defined(MVLM_CONV_TIMING)  // diagnostic build: the tool passes its counter buffer through the environment
    if (const char* e = getenv("MVLM_CONV_TIMING_BUF")) a.timing = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    ConvGrid g;
    if (check_variant<C>(ctx, a, g)) return 1;
    if (a.kparts > 1) {
        MVLM_REQUIRE(ctx, g.tiles <= MVLM_KPARTS_MAX_TILES && g.nblk <= MVLM_KPARTS_MAX_PARTS, "conv: too many tiles for the K-part workspace");
        if (mvlm_conv_kparts_workspace(ctx, &a.kws, &a.kcnt)) return 1;
    }
    if (set_variant_attributes<C>(ctx, variant_id)) return 1;
    if (a.amax_val) {
        if constexpr (C::HAS_AMAX) {
            hipLaunchKernelGGL((conv_mfma_kernel<C, true>), dim3((unsigned)g.nblk), dim3(256), C::LDS_BYTES, ctx->cur_stream(),
                               a, g.tiles_x, g.tiles_y, g.cout_tiles);
        } else {
            return ctx->fail("conv: fused argmax is only built for the 8x32-pixel tile variants");
        }
    } else if (a.in2) {
        if constexpr (C::HAS_IN2) {
            hipLaunchKernelGGL((conv_mfma_kernel<C, false, true>), dim3((unsigned)g.nblk), dim3(256), C::LDS_BYTES, ctx->cur_stream(), a,
                               g.tiles_x, g.tiles_y, g.cout_tiles);
        } else {
            return ctx->fail("conv: this kernel variant has no second-input form");
        }
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<C, false>), dim3((unsigned)g.nblk), dim3(256), C::LDS_BYTES, ctx->cur_stream(), a,
                           g.tiles_x, g.tiles_y, g.cout_tiles);
    }
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

// two independent convolutions on this variant's tiles in one grid (conv_pair_kernel); kparts of each problem in its ConvArgs
template <class C>
int launch_variant_pair(mvlm_ctx* ctx, const ConvArgs& a0, const ConvArgs& a1, int variant_id) {
    if constexpr (!C::PAIRABLE) {
        return ctx->fail("conv: this kernel variant has no two-problem form");
    } else {
        ConvPairArgs p;
        p.a[0] = a0;
        p.a[1] = a1;
        ConvGrid g[2];
        for (int i = 0; i < 2; ++i) {
            ConvArgs& a = p.a[i];
            MVLM_REQUIRE(ctx, !a.amax_val && a.n_par == 1 && !a.timing && !a.in2, "conv: a paired launch takes plain convolutions");
            if (check_variant<C>(ctx, a, g[i])) return 1;
        }
        if (p.a[0].kparts > 1 || p.a[1].kparts > 1) {
            // one workspace per launch stream: problem 1's partial tiles and counters follow problem 0's
            const long t0 = p.a[0].kparts > 1 ? g[0].tiles : 0, n0 = p.a[0].kparts > 1 ? g[0].nblk : 0;
            const long t1 = p.a[1].kparts > 1 ? g[1].tiles : 0, n1 = p.a[1].kparts > 1 ? g[1].nblk : 0;
            MVLM_REQUIRE(ctx, t0 + t1 <= MVLM_KPARTS_MAX_TILES && n0 + n1 <= MVLM_KPARTS_MAX_PARTS, "conv: too many tiles for the K-part workspace");
            float* kws = nullptr;
            unsigned* kcnt = nullptr;
            if (mvlm_conv_kparts_workspace(ctx, &kws, &kcnt)) return 1;
            p.a[0].kws = kws;
            p.a[0].kcnt = kcnt;
            p.a[1].kws = kws + size_t(n0) * 1024;
            p.a[1].kcnt = kcnt + t0;
        }
        for (int i = 0; i < 2; ++i) {
            p.tiles_x[i] = g[i].tiles_x;
            p.tiles_y[i] = g[i].tiles_y;
            p.cout_tiles[i] = g[i].cout_tiles;
            p.nblk[i] = int(g[i].nblk);
        }
        p.nblk0_pad = (p.nblk[0] + 7) / 8 * 8;
        const long total = long(p.nblk0_pad) + p.nblk[1];
        MVLM_REQUIRE(ctx, total < (1l << 31), "conv: bad grid");
        if (set_variant_attributes<C>(ctx, variant_id)) return 1;
        hipLaunchKernelGGL((conv_pair_kernel<C>), dim3((unsigned)total), dim3(256), C::LDS_BYTES, ctx->cur_stream(), p);
        MVLM_CHECK_HIP(ctx, hipGetLastError());
        return 0;
    }
}


}  // namespace
#endif
