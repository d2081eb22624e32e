// Cost of a software grid barrier on MI355X: <blocks> workgroups of 256 threads, one monotonic counter in device memory
// (release fence + atomicAdd by thread 0, bounded spin on a volatile load, acquire fence).  What a persistent kernel over
// the small hourglass levels would pay per layer instead of a kernel launch.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_loop(unsigned* cnt, int rounds, unsigned* fail, float* sink) {
    const unsigned nblk = gridDim.x;
    float acc = float(threadIdx.x);
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.f;  // a little work
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(cnt, 1u);
            const unsigned target = unsigned(r + 1) * nblk;
            long spins = 0;
            while (*(volatile unsigned*)cnt < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { *fail = 1u; break; }
            }
            __threadfence();
        }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main() {
    unsigned *cnt, *fail;
    float* sink;
    hipMalloc(&cnt, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4);
    for (int blocks : {64, 128, 256, 512}) {
        for (int rounds : {200, 2000}) {
            hipMemset(cnt, 0, 4); hipMemset(fail, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(barrier_loop, dim3(blocks), dim3(256), 0, 0, cnt, rounds, fail, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned f = 0; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("blocks %4d rounds %5d: %.3f ms total, %.2f us per barrier%s\n", blocks, rounds, ms, 1e3 * ms / rounds, f ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    return 0;
}
