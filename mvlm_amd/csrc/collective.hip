// The one exchange of the sharded path (the reference's own multi-GPU mechanism is the dormant nn.DataParallel of
// prediction/paulsenpredictor.py:100-105): every rank has the heatmap maxima of ITS views, the per-landmark consensus needs
// all views in pose-table order.  mvlm_allgather_maxima is that exchange for a host that holds an RCCL communicator itself
// (a C / C++ host; the Python host goes through torch.distributed, mvlm_amd/parallel.py, whose layout this mirrors):
//   pack    local  f32[NL, n_local, 3]  ->  send f32[n_max, NL, 3]   (n_max = ceil(N / world); short shards zero-padded)
//   ncclAllGather (RCCL over xGMI) on the context's stream          ->  recv f32[world, n_max, NL, 3]
//   unpack  recv   ->  all f32[NL, N, 3], view v taken from its owner's slot (contiguous shards, sizes differing by <= 1)
// RCCL is not linked: the entry point is looked up at first use - MVLM_RCCL_LIB (a path) if set, else whatever RCCL the
// process has loaded already (the library the caller's communicator came from), else librccl.so.1 - so the library loads
// on machines without RCCL and single-GPU users never touch it.
#include <dlfcn.h>

#include "common.h"

namespace {

using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
using ErrStrFn = const char* (*)(int);
constexpr int NCCL_FLOAT = 7;  // ncclFloat32 (rccl.h ncclDataType_t)

struct Rccl {
    AllGatherFn all_gather = nullptr;
    ErrStrFn err = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        if (const char* p = getenv("MVLM_RCCL_LIB")) {
            h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                x.why = std::string("MVLM_RCCL_LIB: ") + dlerror();
                return x;
            }
        }
        void* f = h ? dlsym(h, "ncclAllGather") : dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!f && !h) {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            if (h) f = dlsym(h, "ncclAllGather");
        }
        if (!f) {
            x.why = "no RCCL in this process (ncclAllGather not found; MVLM_RCCL_LIB names the library)";
            return x;
        }
        x.all_gather = reinterpret_cast<AllGatherFn>(f);
        x.err = reinterpret_cast<ErrStrFn>(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
        return x;
    }();
    return r;
}

// contiguous shard of rank r: [start, start + size) (mvlm_amd/parallel.py shard_range)
__host__ __device__ inline void shard(int n, int world, int r, int* start, int* size) {
    const int base = n / world, rem = n % world;
    *start = r * base + (r < rem ? r : rem);
    *size = base + (r < rem ? 1 : 0);
}

__global__ void pack_kernel(const float* __restrict__ local, int nl, int n_local, int n_max, float* __restrict__ send) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // element of send [n_max, NL, 3]
    if (i >= n_max * nl * 3) return;
    const int c = i % 3, lm = (i / 3) % nl, v = i / (3 * nl);
    send[i] = v < n_local ? local[(size_t(lm) * n_local + v) * 3 + c] : 0.0f;
}

__global__ void unpack_kernel(const float* __restrict__ recv, int nl, int n_total, int n_max, int world, float* __restrict__ all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // element of all [NL, N, 3]
    if (i >= nl * n_total * 3) return;
    const int c = i % 3, v = (i / 3) % n_total, lm = i / (3 * n_total);
    const int base = n_total / world, rem = n_total % world;
    // owner of view v: the first `rem` ranks hold base + 1 views
    const int r = v < rem * (base + 1) ? v / (base + 1) : rem + (v - rem * (base + 1)) / (base > 0 ? base : 1);
    int start, size;
    shard(n_total, world, r, &start, &size);
    all[i] = recv[((size_t(r) * n_max + (v - start)) * nl + lm) * 3 + c];
}

}  // namespace

// The two halves around the transport, for a host that moves the slots itself (MPI, hipMemcpyPeerAsync to one rank - SURVEY.md 8e
// names that as equivalent): slot layout f32[n_max, NL, 3] per rank, n_max = ceil(n_total / world).
extern "C" int mvlm_gather_pack(mvlm_ctx* ctx, const float* maxima_local_dev, int n_local, int n_max, int n_landmarks, float* slot_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, slot_dev && n_max > 0 && n_landmarks > 0 && n_local >= 0 && n_local <= n_max && (maxima_local_dev || n_local == 0),
                 "gather_pack: bad arguments");
    const size_t slot = size_t(n_max) * n_landmarks * 3;
    hipLaunchKernelGGL(pack_kernel, dim3(unsigned((slot + 255) / 256)), dim3(256), 0, ctx->stream, maxima_local_dev, n_landmarks, n_local,
                       n_max, slot_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_gather_unpack(mvlm_ctx* ctx, const float* slots_dev, int world, int n_total, int n_landmarks, float* maxima_all_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, slots_dev && maxima_all_dev && world >= 1 && n_total > 0 && n_landmarks > 0, "gather_unpack: bad arguments");
    const int n_max = (n_total + world - 1) / world;
    const size_t total = size_t(n_landmarks) * n_total * 3;
    hipLaunchKernelGGL(unpack_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, ctx->stream, slots_dev, n_landmarks, n_total, n_max,
                       world, maxima_all_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_allgather_maxima(mvlm_ctx* ctx, void* nccl_comm, int rank, int world, const float* maxima_local_dev,
                                     int n_total, int n_landmarks, float* maxima_all_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "allgather_maxima: bad rank / world");
    MVLM_REQUIRE(ctx, n_total > 0 && n_landmarks > 0 && maxima_all_dev, "allgather_maxima: empty problem");
    MVLM_REQUIRE(ctx, nccl_comm || world == 1, "allgather_maxima: no communicator (only a world of one can do without)");
    int start, n_local;
    shard(n_total, world, rank, &start, &n_local);
    MVLM_REQUIRE(ctx, maxima_local_dev || n_local == 0, "allgather_maxima: this rank holds views but passed no maxima");
    const int n_max = (n_total + world - 1) / world;
    const size_t slot = size_t(n_max) * n_landmarks * 3;
    auto* send = static_cast<float*>(ctx->get_scratch("gather.send", slot * sizeof(float)));
    auto* recv = static_cast<float*>(ctx->get_scratch("gather.recv", slot * world * sizeof(float)));
    MVLM_REQUIRE(ctx, send && recv, "allgather_maxima: scratch allocation failed");
    hipLaunchKernelGGL(pack_kernel, dim3(unsigned((slot + 255) / 256)), dim3(256), 0, ctx->stream, maxima_local_dev, n_landmarks,
                       n_local, n_max, send);
    if (nccl_comm) {
        Rccl& r = rccl();
        if (!r.all_gather) return ctx->fail("allgather_maxima: " + r.why);
        const int rc = r.all_gather(send, recv, slot, NCCL_FLOAT, nccl_comm, ctx->stream);
        if (rc != 0) return ctx->fail(std::string("allgather_maxima: ncclAllGather failed: ") + (r.err ? r.err(rc) : "?"));
    } else {
        MVLM_CHECK_HIP(ctx, hipMemcpyAsync(recv, send, slot * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    const size_t total = size_t(n_landmarks) * n_total * 3;
    hipLaunchKernelGGL(unpack_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, ctx->stream, recv, n_landmarks, n_total, n_max,
                       world, maxima_all_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}
