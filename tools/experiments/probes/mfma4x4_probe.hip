// Layout and rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 blocks per instruction): which lane feeds which
// row / column, where the results land, and cycles per instruction next to v_mfma_f32_32x32x2_f32.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/probes/mfma4x4_probe.hip -o /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void rate(float* out, int iters) {
    const int l = threadIdx.x;
    float x = float(l) * 1e-3f, y = float(l & 7) * 1e-3f;
    f32x4 c4[4] = {};
    f32x16 c16[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) c4[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, c4[k & 3], 0, 0, 0);
            else c16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c16[k & 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 4; ++v) s += c4[i][v];
    for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) s += c16[i][v];
    out[blockIdx.x * 256 + l] = s;
}

int main() {
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 * 256 * 4);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = float(1 + l); hb[l] = float(100 + l); }  // A_l * B_m identifies (l, m)
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int h1 = 0, h2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const int blk = l / 4;
            h1 += hd[l * 4 + v] == ha[4 * blk + v] * hb[l];   // VGPR v = row v (A of lane 4 blk + v), column = own B
            h2 += hd[l * 4 + v] == ha[l] * hb[4 * blk + v];   // VGPR v = column v, row = own A
        }
    printf("layout: VGPR v holds A[lane 4b+v] * B[own lane]: %d / 256;  A[own lane] * B[lane 4b+v]: %d / 256\n", h1, h2);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    for (int kind = 0; kind < 2; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 256;
        if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, 10); else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double n = double(iters) * 16;   // instructions per wave
        const double macs = kind == 0 ? 256.0 : 2048.0;
        printf("%s: %.3f ms, %.1f ns per instruction per wave, %.1f TFLOP/s over %d CUs x 4 waves\n", kind == 0 ? "4x4x1_16b" : "32x32x2", ms,
               ms * 1e6 / n, 2.0 * macs * n * blocks * 4 / (ms * 1e-3) / 1e12, blocks);
    }
    return 0;
}
