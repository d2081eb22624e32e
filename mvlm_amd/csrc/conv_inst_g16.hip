// Instantiates group 16 of the convolution kernel variants (conv_variants.h).
#include "conv_kernel.h"
#include "conv_variants.h"

#define X(id, name, ...) \
    int mvlm_conv_launch_##id(mvlm_ctx* ctx, const ConvArgs& a) { return launch_variant<__VA_ARGS__>(ctx, a, id); } \
    int mvlm_conv_pair_launch_##id(mvlm_ctx* ctx, const ConvArgs& a0, const ConvArgs& a1) { return launch_variant_pair<__VA_ARGS__>(ctx, a0, a1, id); }
MVLM_CONV_VARIANTS_G16(X)
#undef X
