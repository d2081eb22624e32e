// Micro-benchmark: do exact-f32 MFMAs (v_mfma_f32_32x32x2_f32) and f32 VALU FMAs (v_fma_f32) of ONE wave's
// instruction stream overlap on gfx950, i.e. could a convolution kernel add VALU FMAs on top of its matrix
// pipe work?  For V = 0..32 v_fma_f32 placed after every MFMA it reports the kernel time, the MFMA rate and
// the combined (MFMA + VALU) f32 rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/probes/coexec_probe.hip -o /tmp/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

// NM independent MFMA accumulators, V VALU FMAs (independent accumulators, scalar multiplier) after each MFMA
template <int NM, int V, bool MFMA_ON>
__global__ __launch_bounds__(256, 2) void probe(const float* in, float* out, int iters, float wscalar) {
    const int tid = threadIdx.x;
    f32x16 acc[NM];
    for (int m = 0; m < NM; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = in[tid], b = in[tid + 256];
    float vacc[V > 0 ? V : 1];
    for (int v = 0; v < (V > 0 ? V : 1); ++v) vacc[v] = in[tid + v];
    float x = in[tid + 7];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (MFMA_ON) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vacc[v]) : "s"(wscalar), "v"(x));
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < NM; ++m)
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int v = 0; v < V; ++v) s += vacc[v];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NM, int V, bool MFMA_ON>
void run(const float* in, float* out, int blocks) {
    const int iters = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NM, V, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, in, out, iters, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NM, V, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, in, out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = double(blocks) * 4 * iters * 8 * NM * (MFMA_ON ? 1 : 0);
    const double n_fma = double(blocks) * 4 * iters * 8 * NM * V;
    const double tf_m = n_mfma * 4096.0 / (ms * 1e-3) / 1e12, tf_v = n_fma * 128.0 / (ms * 1e-3) / 1e12;
    printf("blocks %5d  NM %d  V %2d  mfma %d  %8.3f ms   MFMA %7.2f  VALU %7.2f  sum %7.2f TFLOP/s\n", blocks, NM, V,
           int(MFMA_ON), ms, tf_m, tf_v, tf_m + tf_v);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = float((i * 37) % 101) / 101.f - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 2048}) {  // 1 and 2 workgroups (= waves per SIMD) per CU, and a long grid
        run<4, 0, true>(in, out, blocks);
        run<4, 4, true>(in, out, blocks);
        run<4, 8, true>(in, out, blocks);
        run<4, 12, true>(in, out, blocks);
        run<4, 16, true>(in, out, blocks);
        run<4, 24, true>(in, out, blocks);
        run<4, 32, true>(in, out, blocks);
        run<4, 16, false>(in, out, blocks);
        run<4, 32, false>(in, out, blocks);
    }
    return 0;
}
