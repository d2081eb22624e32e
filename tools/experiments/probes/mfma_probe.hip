// Micro-benchmark: what limits a v_mfma_f32_32x32x2_f32 stream fed from LDS on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/probes/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

// MODE 0: operands in registers, MT x NT independent accumulators
// MODE 1: operands re-read from LDS every k-step (ds_read, prefetched one step ahead)
// MODE 2: MODE 1 + __syncthreads every 36 k-steps
template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256, 1) void probe(const float* in, float* out, int iters) {
    __shared__ float lds[12288];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < 12288; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 acc[MT][NT];
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float av[2][MT], bv[2][NT];
    for (int m = 0; m < MT; ++m) av[0][m] = lds[half * 128 + m * 32 + l31];
    for (int n = 0; n < NT; ++n) bv[0][n] = lds[9216 + half * 340 + n * 34 + l31];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            if (MODE >= 1) {
                const int nx = (ks + 1) % 36;
#pragma unroll
                for (int m = 0; m < MT; ++m) av[(ks + 1) & 1][m] = lds[nx * 256 + half * 128 + m * 32 + l31];
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[(ks + 1) & 1][n] = lds[9216 + (nx & 7) * 340 + half * 340 + n * 34 + l31 + (nx >> 3)];
            }
            const int cur = MODE >= 1 ? (ks & 1) : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m], bv[cur][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int MT, int NT>
void run(const char* name, const float* in, float* out, int blocks) {
    const int iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, MT, NT>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, MT, NT>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = double(blocks) * 4 * iters * 36 * MT * NT * 4096.0;
    printf("%-28s blocks %5d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 12288 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(12288);
    for (int i = 0; i < 12288; ++i) h[i] = float((i * 37) % 101) / 101.f - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 1024}) {
        run<0, 4, 2>("regs 4x2", in, out, blocks);
        run<1, 4, 2>("lds 4x2", in, out, blocks);
        run<2, 4, 2>("lds+barrier 4x2", in, out, blocks);
        run<0, 2, 4>("regs 2x4", in, out, blocks);
        run<1, 2, 4>("lds 2x4", in, out, blocks);
        run<0, 2, 2>("regs 2x2", in, out, blocks);
        run<1, 2, 2>("lds 2x2", in, out, blocks);
        run<0, 1, 1>("regs 1x1", in, out, blocks);
    }
    return 0;
}
