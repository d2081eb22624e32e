// Instantiates group 14 of the convolution kernel variants (conv_variants.h).
#include "conv_kernel.h"
#include "conv_variants.h"

#define X(id, name, ...) \
    int mvlm_conv_launch_##id(mvlm_ctx* ctx, const ConvArgs& a) { return launch_variant<__VA_ARGS__>(ctx, a, id); }
MVLM_CONV_VARIANTS_G14(X)
#undef X
