// Small memory-bound kernels around the convolution stack: input plane selection,
// 2x2 max-pool, heatmap maxima.
#include "common.h"

namespace {

// numpy's pairwise summation for n = 31 (< 128 block): eight running partial sums over
// the first 24 elements, a fixed combine tree, then the 7-element tail added in order.
template <class T, class F>
__device__ inline T np_sum31(F get) {
    T r[8];
    for (int k = 0; k < 8; ++k) r[k] = get(k);
    for (int i = 8; i < 24; i += 8)
        for (int k = 0; k < 8; ++k) r[k] += get(i + k);
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int i = 24; i < 31; ++i) res += get(i);
    return res;
}

// [N,256,256,4] f32 (renderer layout, render3d.py:117) -> planar [N,C,256,256] with the
// planes the model was built for (BHWC -> BCHW, paulsenpredictor.py:184-185).
__global__ void pack_input_kernel(const float4* __restrict__ img, int n_pix_total, int hw, int4 sel, int c,
                                  float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix_total) return;
    const float4 v = img[i];
    const float px[4] = {v.x, v.y, v.z, v.w};
    const int b = i / hw, p = i - b * hw;
    const int s[4] = {sel.x, sel.y, sel.z, sel.w};
    for (int k = 0; k < c; ++k) out[(size_t(b) * c + k) * hw + p] = px[s[k]];
}

// F.max_pool2d(x, 2) over planar tensors (paulsenpredictor.py:308,313,318,323,328,411)
__global__ void maxpool2_kernel(const float* __restrict__ in, long n_out, int Wo, float* __restrict__ out) {
    const long i = (long(blockIdx.x) * blockDim.x + threadIdx.x) * 2;  // two outputs per thread
    if (i >= n_out) return;
    const long row = i / Wo;          // plane*Ho + yo
    const int xo = int(i - row * Wo);
    const float* src = in + row * 2 * (2l * Wo) + 2 * xo;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 2 * Wo);
    float2 o;
    o.x = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
    o.y = fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w));
    *reinterpret_cast<float2*>(out + i) = o;
}

// reduce the conv epilogue's per-tile argmax partials to (row-1, col-0.5, value)
// (paulsenpredictor.py:123-127): first maximum in row-major order.
__global__ void amax_final_kernel(const float* __restrict__ val, const int* __restrict__ idx, int nl, int parts,
                                  int size, int view0, int n_views_total, float* __restrict__ maxima, int* __restrict__ best_idx) {
    const int b = blockIdx.x / nl, lm = blockIdx.x % nl;
    const float* v = val + size_t(blockIdx.x) * parts;
    const int* ix = idx + size_t(blockIdx.x) * parts;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < parts; i += blockDim.x) {
        const float ov = v[i];
        const int oi = ix[i];
        if (argmax_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(bv, s);
        const int oi = __shfl_xor(bi, s);
        if (argmax_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < int(blockDim.x >> 6); ++w)
            if (argmax_better(sv[w], si[w], bv, bi)) {
                bv = sv[w];
                bi = si[w];
            }
        float* o = maxima + (size_t(lm) * n_views_total + view0 + b) * 3;
        o[0] = float(bi / size) - 1.f;
        o[1] = float(bi % size) - 0.5f;
        o[2] = bv;
        if (best_idx) best_idx[blockIdx.x] = bi;  // flat pixel index of the maximum (the moment refinement's window centre)
    }
}

// maxima of materialised heatmaps, "simple" and "moment" (paulsenpredictor.py:112-158).
// One workgroup per (view, landmark) plane.
__global__ void heatmap_maxima_kernel(const float* __restrict__ heat, int n_views, int nl, int size, int method,
                                      float* __restrict__ out) {
    const int b = blockIdx.x / nl, lm = blockIdx.x % nl;
    const float* hm = heat + size_t(blockIdx.x) * size * size;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
        const float ov = hm[i];
        if (argmax_better(ov, i, bv, bi)) {
            bv = ov;
            bi = i;
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(bv, s);
        const int oi = __shfl_xor(bi, s);
        if (argmax_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float rows[31], cols[31];
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    for (int w = 0; w < int(blockDim.x >> 6); ++w)
        if (argmax_better(sv[w], si[w], bv, bi)) {
            bv = sv[w];
            bi = si[w];
        }
    const int px = bi / size, py = bi % size;
    float fx = float(px), fy = float(py);
    constexpr int sz = 15;
    const bool refine = method == MVLM_MAXIMA_MOMENT && px > sz && size - px > sz && py > sz && size - py > sz;
    if (refine) {
        // 31x31 window centroid (paulsenpredictor.py:142-154).  np.sum(axis=1) reduces the
        // contiguous axis with numpy's pairwise routine, np.sum(axis=0) adds row after row;
        // the weighted sums run in float64 because np.arange is int64.
        if (threadIdx.x < 31) {
            const float* rowp = hm + (px - sz + threadIdx.x) * size + (py - sz);
            rows[threadIdx.x] = np_sum31<float>([&](int j) { return rowp[j]; });
        } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 31) {
            const int t = threadIdx.x - 64;
            float s = 0.f;
            for (int j = 0; j < 31; ++j) s += hm[(px - sz + j) * size + (py - sz + t)];
            cols[t] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (refine) {
            double s = np_sum31<double>([&](int j) { return double(j) * double(rows[j]); });
            float ss = np_sum31<float>([&](int j) { return rows[j]; });
            const double posx = s / double(ss) - sz;
            s = np_sum31<double>([&](int j) { return double(j) * double(cols[j]); });
            ss = np_sum31<float>([&](int j) { return cols[j]; });
            const double posy = s / double(ss) - sz;
            // coordinates[k] is float32: (px + pos) - 1 evaluated in float64, then stored
            fx = float((double(px) + posx) - 1.0);
            fy = float((double(py) + posy) - 0.5);
        } else {
            fx = float(px) - 1.f;
            fy = float(py) - 0.5f;
        }
        float* o = out + (size_t(lm) * n_views + b) * 3;
        o[0] = fx;
        o[1] = fy;
        o[2] = bv;
    }
}


// "moment" selection without the [N,NL,256,256] heatmaps (paulsenpredictor.py:129-156).  The fused argmax has found the
// peak of every (view, landmark) plane; where the reference refines it (peak more than 15 pixels from every border) this
// kernel RECOMPUTES the 31x31 heatmap window around it from conv10's low-resolution output - the four parity forms of
// conv11 (2x2 kernels over the 128x128 tensor, weights.collapse_upsampled_3x3) - and takes the centroid in numpy's order.
// The window values are bit for bit those of the convolution kernel (conv2x2_c*_t8x32), whose f32 MFMAs are a k-ordered
// fmaf chain (cdna_hip_programming.md, "FP32-input MFMA"): per output, chunks of 4 input channels, inside a chunk the four
// taps in row-major order, inside a tap the chunk's channels - the 32-row tiles, the 16-row and the 4-row strip alike -
// then + bias.  One workgroup per (view, landmark); a thread owns up to four window pixels; a chunk's 4 x 18 x 18 patch of
// the low-resolution tensor is staged in LDS.  Planes whose peak is near a border keep the simple form amax_final wrote.
struct MomentParityWeights {
    const float* w[4];  // [tap][cin_pad][cout_pad] of parity 2 * (row & 1) + (col & 1)
};

__global__ __launch_bounds__(256) void moment_refine_kernel(const float* __restrict__ x10, int nl, int cin_pad, int cout_pad,
                                                            const MomentParityWeights wp, const float* __restrict__ bias,
                                                            const int* __restrict__ best_idx, int view0, int n_views_total,
                                                            float* __restrict__ maxima) {
    constexpr int SIZE = 256, LOW = 128, SZ = 15, WIN = 31, PATCH = 18;
    const int b = blockIdx.x / nl, lm = blockIdx.x % nl;
    const int bi = best_idx[blockIdx.x];
    const int px = bi / SIZE, py = bi % SIZE;
    if (!(px > SZ && SIZE - px > SZ && py > SZ && SIZE - py > SZ)) return;  // (uniform per workgroup)
    extern __shared__ float smem_m[];
    float* const sw = smem_m;                               // [parity][tap][cin_pad]: this landmark's weights
    float* const sx = sw + 16 * cin_pad;                    // [4][PATCH][PATCH]
    float* const win = sx + 4 * PATCH * PATCH;              // [WIN][WIN]
    float* const rows = win + WIN * WIN;                    // [WIN], then cols [WIN]
    float* const cols = rows + WIN;
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * cin_pad; e += 256) {
        const int par = e / (4 * cin_pad), r = e - par * 4 * cin_pad;  // r = tap * cin_pad + channel
        sw[e] = wp.w[par][size_t(r) * cout_pad + lm];
    }
    const int Y0 = px - SZ, X0 = py - SZ;
    const int ly0 = (Y0 >> 1) - 1, lx0 = (X0 >> 1) - 1;     // low-resolution origin of the staged patch
    int off[4], wbase[4];
    bool ok[4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = tid + 256 * j;
        ok[j] = o < WIN * WIN;
        const int wy = ok[j] ? o / WIN : 0, wx = ok[j] ? o % WIN : 0;
        const int Y = Y0 + wy, X = X0 + wx;
        const int a = Y & 1, c = X & 1;
        off[j] = ((Y >> 1) + a - 1 - ly0) * PATCH + ((X >> 1) + c - 1 - lx0);  // top-left of the pixel's 2x2 window in the patch
        wbase[j] = (2 * a + c) * 4 * cin_pad;
    }
    const float* const xb = x10 + size_t(b) * nl * LOW * LOW;
    for (int cb = 0; cb < cin_pad; cb += 4) {
        __syncthreads();  // the previous chunk's patch has been read
        for (int e = tid; e < 4 * PATCH * PATCH; e += 256) {
            const int ch = e / (PATCH * PATCH), r = e - ch * PATCH * PATCH;
            const int gy = ly0 + r / PATCH, gx = lx0 + r % PATCH;
            const bool in = cb + ch < nl && gy >= 0 && gy < LOW && gx >= 0 && gx < LOW;
            sx[e] = in ? xb[(size_t(cb + ch) * LOW + gy) * LOW + gx] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!ok[j]) continue;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    acc[j] = fmaf(sw[wbase[j] + tap * cin_pad + cb + ch], sx[ch * PATCH * PATCH + off[j] + (tap >> 1) * PATCH + (tap & 1)], acc[j]);
        }
    }
    const float bv = bias ? bias[lm] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ok[j]) win[tid + 256 * j] = acc[j] + bv;
    __syncthreads();
    // the centroid, as in heatmap_maxima_kernel: np.sum(axis=1) = numpy's pairwise routine over the contiguous axis,
    // np.sum(axis=0) adds row after row, the weighted sums run in float64 (np.arange is int64)
    if (tid < WIN) {
        const float* rowp = win + tid * WIN;
        rows[tid] = np_sum31<float>([&](int j) { return rowp[j]; });
    } else if (tid >= 64 && tid < 64 + WIN) {
        const int t = tid - 64;
        float s = 0.f;
        for (int j = 0; j < WIN; ++j) s += win[j * WIN + t];
        cols[t] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double s = np_sum31<double>([&](int j) { return double(j) * double(rows[j]); });
        float ss = np_sum31<float>([&](int j) { return rows[j]; });
        const double posx = s / double(ss) - SZ;
        s = np_sum31<double>([&](int j) { return double(j) * double(cols[j]); });
        ss = np_sum31<float>([&](int j) { return cols[j]; });
        const double posy = s / double(ss) - SZ;
        float* o = maxima + (size_t(lm) * n_views_total + view0 + b) * 3;
        o[0] = float((double(px) + posx) - 1.0);
        o[1] = float((double(py) + posy) - 0.5);
    }
}

}  // namespace

int mvlm_launch_pack_input(mvlm_ctx* ctx, const float* images, int n, const int* sel4, int c, float* out) {
    const int hw = MVLM_IMAGE_SIZE * MVLM_IMAGE_SIZE;
    const int total = n * hw;
    hipLaunchKernelGGL(pack_input_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->cur_stream(),
                       reinterpret_cast<const float4*>(images), total, hw, make_int4(sel4[0], sel4[1], sel4[2], sel4[3]),
                       c, out);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

int mvlm_launch_maxpool2(mvlm_ctx* ctx, const float* in, int planes, int H, int W, float* out) {
    MVLM_REQUIRE(ctx, W % 4 == 0 && H % 2 == 0, "maxpool: size must be a multiple of 4");
    const long n_out = long(planes) * (H / 2) * (W / 2);
    const long threads = n_out / 2;
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->cur_stream(), in, n_out,
                       W / 2, out);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

int mvlm_launch_amax_final(mvlm_ctx* ctx, const float* val, const int* idx, int n_img, int view0, int n_views_total,
                           int nl, int parts, int size, float* maxima, int* best_idx) {
    hipLaunchKernelGGL(amax_final_kernel, dim3(n_img * nl), dim3(256), 0, ctx->cur_stream(), val, idx, nl, parts, size, view0,
                       n_views_total, maxima, best_idx);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

namespace {
__global__ void fp16_poison_kernel(const unsigned* __restrict__ flag, float* __restrict__ maxima, int n_img, int view0, int n_views_total,
                                   int nl, float* __restrict__ heat) {
    if (*flag == 0u) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (image, landmark)
    if (i >= n_img * nl) return;
    const int b = i / nl, lm = i % nl;
    if (maxima) maxima[(size_t(lm) * n_views_total + view0 + b) * 3 + 2] = __builtin_nanf("");
    if (heat) heat[size_t(i) * MVLM_IMAGE_SIZE * MVLM_IMAGE_SIZE] = __builtin_nanf("");
}
}  // namespace

int mvlm_launch_fp16_poison(mvlm_ctx* ctx, const unsigned* flag, float* maxima, int n_img, int view0, int n_views_total, int nl,
                            float* heat) {
    MVLM_REQUIRE(ctx, flag && (maxima || heat), "fp16 poison: null argument");
    hipLaunchKernelGGL(fp16_poison_kernel, dim3((n_img * nl + 255) / 256), dim3(256), 0, ctx->cur_stream(), flag, maxima, n_img, view0,
                       n_views_total, nl, heat);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

int mvlm_launch_moment_refine(mvlm_ctx* ctx, const float* x10, int n_img, int nl, int cin_pad, int cout_pad, const float* const w_par[4],
                              const float* bias, const int* best_idx, int view0, int n_views_total, float* maxima) {
    MVLM_REQUIRE(ctx, x10 && best_idx && maxima && w_par[0] && w_par[1] && w_par[2] && w_par[3], "moment: null argument");
    MVLM_REQUIRE(ctx, cin_pad % 4 == 0 && cin_pad >= nl && cout_pad >= nl, "moment: bad channel padding");
    MomentParityWeights wp;
    for (int q = 0; q < 4; ++q) wp.w[q] = w_par[q];
    const size_t lds = size_t(16 * cin_pad + 4 * 18 * 18 + 31 * 31 + 2 * 31) * sizeof(float);
    MVLM_REQUIRE(ctx, lds <= 64 * 1024, "moment: too many input channels for the weight table in LDS");
    hipLaunchKernelGGL(moment_refine_kernel, dim3(n_img * nl), dim3(256), lds, ctx->cur_stream(), x10, nl, cin_pad, cout_pad, wp, bias,
                       best_idx, view0, n_views_total, maxima);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_heatmap_maxima(mvlm_ctx* ctx, const float* heat_dev, int n_views, int n_landmarks, int size,
                                   int method, float* out_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, heat_dev && out_dev && n_views > 0 && n_landmarks > 0 && size > 0, "heatmap_maxima: bad arguments");
    MVLM_REQUIRE(ctx, method == MVLM_MAXIMA_SIMPLE || method == MVLM_MAXIMA_MOMENT, "heatmap_maxima: unknown method");
    hipLaunchKernelGGL(heatmap_maxima_kernel, dim3(n_views * n_landmarks), dim3(256), 0, ctx->stream, heat_dev, n_views,
                       n_landmarks, size, method, out_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}
