// The convolution kernel variants: X(id, name, Cfg<COUT_T, TW, TRI, NIMG, KS, CK[, SPLITK]>).
// Grouped so that every group is instantiated in its own translation unit (conv_inst_g*.hip) and
// the library builds in parallel; conv_mfma.hip dispatches through mvlm_conv_launch_<id>().
#ifndef MVLM_CONV_VARIANTS_H
#define MVLM_CONV_VARIANTS_H
#define MVLM_CONV_VARIANTS_G0(X) \
    X(0, "conv3x3_c128_t8x32", Cfg<128, 32, 8, 1, 3, 4>)
#define MVLM_CONV_VARIANTS_G1(X) \
    X(1, "conv3x3_c96_t8x32", Cfg<96, 32, 8, 1, 3, 4>)
#define MVLM_CONV_VARIANTS_G2(X) \
    X(11, "conv2x2_c96_t8x32", Cfg<96, 32, 8, 1, 2, 4>)
#define MVLM_CONV_VARIANTS_G3(X) \
    X(16, "conv3x3_c80_t8x32", Cfg<80, 32, 8, 1, 3, 4>)
#define MVLM_CONV_VARIANTS_G4(X) \
    X(17, "conv2x2_c80_t8x32", Cfg<80, 32, 8, 1, 2, 4>)
#define MVLM_CONV_VARIANTS_G5(X) \
    X(10, "conv3x3_c64_t8x32", Cfg<64, 32, 8, 1, 3, 4>)
#define MVLM_CONV_VARIANTS_G6(X) \
    X(9, "conv3x3_c64_t4x32", Cfg<64, 32, 4, 1, 3, 4>) \
    X(8, "conv3x3_c128_t4x32", Cfg<128, 32, 4, 1, 3, 4>)
#define MVLM_CONV_VARIANTS_G7(X) \
    X(3, "conv3x3_c32_t16x32", Cfg<32, 32, 16, 1, 3, 4>) \
    X(4, "conv1x1_c128_t8x32", Cfg<128, 32, 8, 1, 1, 8>)
#define MVLM_CONV_VARIANTS_G8(X) \
    X(5, "conv3x3_c32_t8x16", Cfg<32, 16, 8, 1, 3, 16>) \
    X(6, "conv3x3_c32_t8x8x2", Cfg<32, 8, 8, 2, 3, 16>) \
    X(7, "conv3x3_c32_t4x4x8", Cfg<32, 4, 4, 8, 3, 16>)
#define MVLM_CONV_VARIANTS_G9(X) \
    X(12, "conv3x3_sk_t2x16", Cfg<32, 16, 2, 1, 3, 32, true>) \
    X(13, "conv3x3_sk_t4x8", Cfg<32, 8, 4, 1, 3, 32, true>) \
    X(14, "conv3x3_sk_t4x4x2", Cfg<32, 4, 4, 2, 3, 32, true>) \
    X(15, "conv3x3_sk_t1x32", Cfg<32, 32, 1, 1, 3, 32, true>)
// split-K tiles with 16- and 8-channel chunks: 49 / 26 KB of LDS instead of 94 KB - three or more workgroups per CU
// and a shorter first-chunk prologue for the latency-bound launches of the small hourglass levels
#define MVLM_CONV_VARIANTS_G10(X) \
    X(18, "conv3x3_sk16_t2x16", Cfg<32, 16, 2, 1, 3, 16, true>) \
    X(19, "conv3x3_sk16_t4x8", Cfg<32, 8, 4, 1, 3, 16, true>) \
    X(20, "conv3x3_sk16_t4x4x2", Cfg<32, 4, 4, 2, 3, 16, true>) \
    X(21, "conv3x3_sk16_t1x32", Cfg<32, 32, 1, 1, 3, 16, true>)
#define MVLM_CONV_VARIANTS_G11(X) \
    X(22, "conv3x3_sk8_t2x16", Cfg<32, 16, 2, 1, 3, 8, true>) \
    X(23, "conv3x3_sk8_t4x8", Cfg<32, 8, 4, 1, 3, 8, true>) \
    X(24, "conv3x3_sk8_t4x4x2", Cfg<32, 4, 4, 2, 3, 8, true>) \
    X(25, "conv3x3_sk8_t1x32", Cfg<32, 32, 1, 1, 3, 8, true>)
// 84 output channels (conv6 / conv10 of the 84-landmark network) = 64 + 16 + 4 rows: no padding to 96
#define MVLM_CONV_VARIANTS_G12(X) \
    X(26, "conv3x3_c84_t8x32", Cfg<84, 32, 8, 1, 3, 4>)
// 8x16-pixel tiles with 8-channel chunks (round 3).  c32k8: the 32 x (8x16) tile in 30 KB of LDS instead of 60 - five
// workgroups per CU instead of two, so the 768 workgroups of a 64x64 level at 12 views are resident at once (no
// half-empty second round): -12..14 % on the 128->64 / 64->64 layers there, -12 % on 256->128 @16x16 at 96 views.
// c64k8: 64 channels on the same pixel tile (halo 10 x 18 instead of 6 x 34 pixels per channel): -4.5 % on 256->128 @64x64
// at 12 views, -5 % on the 32x32 level at 96 views.  Measured and dropped: 128 x (8x16), 64 x (8x16) with 4-channel
// chunks, 32 x (4x32) / 64 x (4x32) with 8-channel chunks, 32 x (8x16) with 4-channel chunks (within 1 % of these),
// a 64 x (16x32) tile (128 accumulators, +0.7 %), input channels divided over workgroups on the plain tiles (slower).
#define MVLM_CONV_VARIANTS_G13(X) \
    X(27, "conv3x3_c32k8_t8x16", Cfg<32, 16, 8, 1, 3, 8>) \
    X(28, "conv3x3_c64k8_t8x16", Cfg<64, 16, 8, 1, 3, 8>)
// conv11's parity convolutions of the 84-landmark network on 84 rows (64 + 16 + 4) instead of 96, fused argmax included
#define MVLM_CONV_VARIANTS_G14(X) \
    X(29, "conv2x2_c84_t8x32", Cfg<84, 32, 8, 1, 2, 4>)
// split-K tiles with TWO 32-pixel columns per workgroup (round 4): the staged weight slice serves twice the matrix work
#define MVLM_CONV_VARIANTS_G15(X) \
    X(30, "conv3x3_sk16_t8x8", Cfg<32, 8, 8, 1, 3, 16, true>) \
    X(31, "conv3x3_sk8_t8x8", Cfg<32, 8, 8, 1, 3, 8, true>) \
    X(32, "conv3x3_sk16_t4x16", Cfg<32, 16, 4, 1, 3, 16, true>)
#define MVLM_CONV_VARIANTS(X) \
    MVLM_CONV_VARIANTS_G0(X) MVLM_CONV_VARIANTS_G1(X) MVLM_CONV_VARIANTS_G2(X) MVLM_CONV_VARIANTS_G3(X) MVLM_CONV_VARIANTS_G4(X) MVLM_CONV_VARIANTS_G5(X) MVLM_CONV_VARIANTS_G6(X) MVLM_CONV_VARIANTS_G7(X) MVLM_CONV_VARIANTS_G8(X) MVLM_CONV_VARIANTS_G9(X) MVLM_CONV_VARIANTS_G10(X) MVLM_CONV_VARIANTS_G11(X) MVLM_CONV_VARIANTS_G12(X) MVLM_CONV_VARIANTS_G13(X) MVLM_CONV_VARIANTS_G14(X) MVLM_CONV_VARIANTS_G15(X)
#define MVLM_CONV_N_GROUPS 16
#endif
