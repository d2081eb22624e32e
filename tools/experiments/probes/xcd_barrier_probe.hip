// Cost of a two-level (XCD-hierarchical) software grid barrier on MI355X, beside the flat one-counter barrier of
// grid_barrier_probe.hip: the workgroups of one group (blockIdx % 8 = the XCD under round-robin placement; any grouping is
// correct, placement only decides the speed) arrive on the group's counter, the group's last arrival goes to the top
// counter, waits for the other groups there and publishes the group's generation word, which the rest of the group polls.
// Every counter sits on a 128-byte line of its own; polls are relaxed agent-scope loads with s_sleep; every spin is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_barrier_probe.hip -o xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>

struct Bar {
    unsigned group_cnt[8][32];
    unsigned gen[8][32];
    unsigned top[32];
    unsigned flat[32];
    unsigned fail[32];
};

__device__ __forceinline__ unsigned poll(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool HIER>
__global__ void barrier_loop(Bar* b, int rounds, float* sink) {
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned g = bid & 7u;
    const unsigned ng = nblk < 8u ? nblk : 8u;             // groups that have members
    const unsigned members = (nblk - g + 7u) / 8u;         // workgroups of this group
    float acc = float(threadIdx.x);
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.f;  // a little work
        __syncthreads();
        if (threadIdx.x == 0) {
            long spins = 0;
            if (HIER) {
                __threadfence();  // release: this workgroup's stores
                const unsigned old = atomicAdd(&b->group_cnt[g][0], 1u);
                if (old == unsigned(r + 1) * members - 1u) {  // the group's last arrival speaks for the group
                    atomicAdd(&b->top[0], 1u);
                    while (poll(&b->top[0]) < unsigned(r + 1) * ng) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 4000000) { b->fail[0] = 1u; break; }
                    }
                    __threadfence();
                    __hip_atomic_store(&b->gen[g][0], unsigned(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (poll(&b->gen[g][0]) < unsigned(r + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 4000000) { b->fail[0] = 1u; break; }
                    }
                }
                __threadfence();  // acquire
            } else {
                __threadfence();
                atomicAdd(&b->flat[0], 1u);
                while (poll(&b->flat[0]) < unsigned(r + 1) * nblk) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000) { b->fail[0] = 1u; break; }
                }
                __threadfence();
            }
        }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main() {
    Bar* bar;
    float* sink;
    hipMalloc(&bar, sizeof(Bar));
    hipMalloc(&sink, 4);
    for (int hier = 0; hier < 2; ++hier)
        for (int blocks : {64, 128, 256, 512}) {
            for (int rounds : {200, 2000}) {
                hipMemset(bar, 0, sizeof(Bar));
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                hipEventRecord(e0);
                if (hier)
                    hipLaunchKernelGGL(barrier_loop<true>, dim3(blocks), dim3(256), 0, 0, bar, rounds, sink);
                else
                    hipLaunchKernelGGL(barrier_loop<false>, dim3(blocks), dim3(256), 0, 0, bar, rounds, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                Bar h;
                hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost);
                printf("%s blocks %4d rounds %5d: %.3f ms total, %.2f us per barrier%s\n", hier ? "two-level" : "flat     ", blocks,
                       rounds, ms, 1e3 * ms / rounds, h.fail[0] ? "  (SPIN LIMIT HIT)" : "");
                fflush(stdout);
            }
        }
    return 0;
}
