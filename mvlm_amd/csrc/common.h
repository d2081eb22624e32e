// Internal declarations shared by the HIP translation units of libmvlm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mvlm_hip.h"
#include "jpeg_plan.h"

#define MVLM_CHECK_HIP(ctx, expr)                                                            \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (ctx)->fail(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

// every entry point: serialise callers of this context and make its GPU the calling thread's
// current device (a process may hold contexts for several GPUs)
#define MVLM_ENTER(ctx)                               \
    std::lock_guard<std::mutex> _mvlm_lock((ctx)->mu); \
    MvlmOrderGuard _mvlm_order{(ctx)};                 \
    MVLM_CHECK_HIP(ctx, hipSetDevice((ctx)->device))

#define MVLM_REQUIRE(ctx, cond, msg)                                                         \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            (ctx)->fail(std::string(msg) + " [" #cond "]");                                  \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------
// np.argmax order on floats (paulsenpredictor.py:123): NaN counts as the maximum, the first one wins; otherwise the
// larger value, the lower index on ties (+0 == -0).  argmax_key maps a float to an int whose integer order is that
// order; argmax_value maps the winning key back (a NaN comes back as a quiet NaN).
__host__ __device__ inline int argmax_key(float v) {
    v = v + 0.0f;  // -0 -> +0
    int b;
    __builtin_memcpy(&b, &v, 4);
    if (v != v) return 0x7fffffff;
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__host__ __device__ inline float argmax_value(int key) {
    int b = key == 0x7fffffff ? 0x7fc00000 : (key >= 0 ? key : (key ^ 0x7fffffff));
    float v;
    __builtin_memcpy(&v, &b, 4);
    return v;
}
// does (ov, oi) precede (bv, bi) in that order?
__host__ __device__ inline bool argmax_better(float ov, int oi, float bv, int bi) {
    const int ok = argmax_key(ov), bk = argmax_key(bv);
    return ok > bk || (ok == bk && oi < bi);
}

// ---------------------------------------------------------------------------------------
// One convolution launch.  All tensors are planar NCHW f32; a tensor argument is a base
// pointer + total channel count + first channel, so a launch can read or write a channel
// slice of a wider tensor (the residual block's concat, paulsenpredictor.py:273).
struct ConvArgs {
    // input
    const float* in = nullptr;
    int in_ctot = 0, in_coff = 0;
    int cin = 0, cin_pad = 0;
    int up_in = 0;  // 1: input tensor is [B][C][H/2][W/2], read through nearest 2x upsampling
    const float* pre_scale = nullptr;  // [cin_pad] BN+ReLU applied to the input; null = raw
    const float* pre_shift = nullptr;
    // geometry (output == conv-input spatial size)
    int B = 0, H = 0, W = 0;
    // weights [taps][cin_pad][cout_pad]
    const float* w = nullptr;
    int cout = 0, cout_pad = 0, ksize = 3;
    const float* bias = nullptr;        // [cout_pad] or null
    const float* post_scale = nullptr;  // [cout_pad] relu(v*s+t) or null
    const float* post_shift = nullptr;
    // residual(s): y = v + (res1 [+ res2])
    const float* res1 = nullptr;
    int res1_ctot = 0, res1_coff = 0;
    const float* res2 = nullptr;
    int res2_ctot = 0, res2_coff = 0;
    // outputs
    float* out_raw = nullptr;  // v (before residual)
    int raw_ctot = 0, raw_coff = 0;
    float* out = nullptr;  // v + residual
    int out_ctot = 0, out_coff = 0;
    // 1: out is [B][out_ctot][2H][2W]; each value is written to its 2x2 block, + skip there
    // 2: out is [B][out_ctot][2H][2W]; the value goes to pixel (2y + sub_y, 2x + sub_x) only
    int up_out = 0;
    int sub_y = 0, sub_x = 0;  // up_out == 2 / ksize == 2: output parity = 2x2 window offset in the halo tile
    // ksize == 2, up_out == 2: all four output parities in one launch (grid x 4): parity pl = 2 sub_y + sub_x uses the
    // weights w_par[pl] and, with a fused argmax, the partials from amax_part0 + pl * amax_par_stride
    int n_par = 1;
    const float* w_par[4] = {nullptr, nullptr, nullptr, nullptr};
    int amax_par_stride = 0;
    const float* skip = nullptr;
    int skip_ctot = 0, skip_coff = 0;
    // consumer-side "upsample x 2 + skip" (round 5; paulsenpredictor.py:334-359 seen from the layer that READS the sum): the
    // input value is in[b][c][y][x] + in2[b][c][y >> 1][x >> 1], in2 = [B][in2_ctot][H/2][W/2] - the hourglass's last block
    // then writes its plain output instead of scattering every value into its 2x2 block of the skip tensor.  Served by the
    // dominant tile only (conv5 / conv9: mvlm_conv_in2_ok).
    const float* in2 = nullptr;
    int in2_ctot = 0;
    // split-K tiles only: input channels divided over `kparts` workgroups per output tile (set by the variant id);
    // kws = partial tiles [tile][part][16][64] f32, kcnt = one arrival counter per tile (zero between launches)
    int kparts = 1;
    float* kws = nullptr;
    unsigned* kcnt = nullptr;
    // fused per-(image, channel) argmax partials (conv11): [B][cout][n_part]
    float* amax_val = nullptr;
    int* amax_idx = nullptr;
    int amax_parts = 0;   // partials per (image, channel) in the buffers
    int amax_part0 = 0;   // first partial this launch writes (parity launches share one buffer)
    // fused 2x2 max-pool of the final values (F.max_pool2d(x, 2, 2), paulsenpredictor.py:304, :413):
    // pool_out is [B][pool_ctot][H/2][W/2]; `out` may be null when only the pooled tensor is consumed
    float* pool_out = nullptr;
    int pool_ctot = 0, pool_coff = 0;
    float fast_unscale = 1.f;  // f16x2 kernel only: inverse of the power-of-two scale its packed weights carry
    unsigned* fast_ovf = nullptr;  // f16x2 kernel only: set to 1 when an activation lies outside fp16's range
    int pool_hint = 0;  // the caller wants the pooled tensor too (set for the variant choice, before pool_out is decided)
    // -DMVLM_CONV_TIMING builds only (tools/conv_phase_timing.py): u64[4] = summed cycles of wave 0 in
    // prologue / K loop / epilogue, number of workgroups
    unsigned long long* timing = nullptr;
};

struct ConvProfileRec {
    int slot, variant;
    double flops;
    hipEvent_t e0, e1;
    short shape[6] = {0, 0, 0, 0, 0, 0};  // ksize, cin_pad, cout_pad, size, kind (conv_kind), batch; slot -1 = the pool kernel
};

// Dispatch key of a convolution launch beyond its shape: 0 plain / residual-block layer, 1 the value is scattered 2x2 into
// the skip tensor (hourglass way up), 2 the 2x2 max-pooled tensor is wanted as well.  The measured tables and the tuning
// overrides are kept per kind: the epilogues differ, and so does the best tile.
struct ConvOverride {
    short ksize, cin_pad, cout_pad, size, kind;
    int variant;
};

// one captured forward pass (hipGraph) of the network for a fixed set of buffers and shapes
struct CnnGraphEntry {
    const void* images = nullptr;
    const void* out = nullptr;   // maxima or heatmap pointer
    const void* ws = nullptr;
    size_t ws_bytes = 0;
    int nb = 0, view0 = 0, n_total = 0, heat = 0;
    int sel4[4] = {0, 0, 0, 0};
    int uses = 0;                // calls seen with this key; the graph is captured on the second one
    hipGraphExec_t exec = nullptr;
};

struct CnnState {
    bool loaded = false;
    int n_landmarks = 0, in_channels = 0;
    float* blob = nullptr;  // device copy of the packed weights
    std::vector<int32_t> desc;  // host copy of the descriptor table
    bool profiling = false;
    std::vector<ConvProfileRec> prof;
    std::vector<hipEvent_t> event_pool;
    size_t event_cursor = 0;
    // second stream + fork / join events of the executor (created on first use), captured graphs
    hipStream_t side_stream = nullptr;
    hipStream_t capture_stream = nullptr;  // passes are recorded here (the caller's stream may be the null stream)
    std::vector<hipEvent_t> sync_events;
    size_t sync_cursor = 0;
    int graph_mode = 1;            // 0: always eager, 1: replay captured graphs when not profiling
    int concurrency = 0;           // 0: one stream (default), 1: lower hourglass pyramid on a side stream (small batches)
    int selection = 0;             // heatmap maxima of mvlm_cnn_maxima: 0 "simple" (argmax pixel), 1 "moment" (31x31 centroid around it)
    int pairing = 1;               // independent residual blocks of a hourglass level share launches: 0 never, 1 where the measured
                                   // table says so (default), 2 wherever one kernel variant can serve both (tests, tuning)
    // opt-in "fast" precision (conv_fast.hip): bf16x3-split weights of the eligible 3x3 layers, per conv slot
    unsigned short* fast_blob = nullptr;
    std::vector<long long> fast_off;   // u16 element offset per slot, -1 = the layer stays on the exact kernel
    int fast = 0;                      // 1: eligible layers run on the bf16x3 kernel, 2: on the f16x2 kernel
    unsigned short* fast16_blob = nullptr;   // the f16x2 form's packed weights, offsets and per-layer inverse scales
    std::vector<long long> fast16_off;
    std::vector<float> fast16_unscale;
    unsigned* fast16_flag = nullptr;         // device word: an f16x2 launch of the current pass met |x| >= 65504
    std::vector<CnnGraphEntry> graphs;
    long graph_replays = 0, graph_captures = 0, eager_runs = 0, graph_failures = 0;
};

struct mvlm_mesh {
    float* verts = nullptr;   // [V,3]
    float* uvs = nullptr;     // [V,2] or null
    int32_t* tris = nullptr;  // [T,3]
    uint8_t* tex = nullptr;   // [H,W,3] or null
    int n_verts = 0, n_tris = 0, tex_h = 0, tex_w = 0;
    unsigned long long uid = 0;  // unique per upload (an address can be recycled): key of per-mesh derived data a context keeps
    size_t cap[4] = {0, 0, 0, 0};  // allocation sizes of verts / uvs / tris / tex (for the ctx's mesh pool)
    // recorded on the context's upload stream behind the four host-to-device copies; every consumer's stream waits for it
    hipEvent_t ready = nullptr;
    hipEvent_t waited[4] = {nullptr, nullptr, nullptr, nullptr};  // "previous owner is done" events of recycled buffers
};

struct RenderProfileRec {
    int n_views, n_verts, n_tris;
    hipEvent_t e0, e1;
};

// per-thread copy of the last error message and the context it belongs to (api.hip: mvlm_last_error)
inline std::string& mvlm_thread_error() {
    thread_local std::string s;
    return s;
}
inline const void*& mvlm_thread_error_ctx() {
    thread_local const void* c = nullptr;
    return c;
}

struct mvlm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // where the network executor's launchers enqueue right now: null = `stream`; the executor points it at
    // its side stream while it issues the branch of the graph that runs beside the main one (cnn_graph.hip)
    hipStream_t launch_stream = nullptr;
    hipStream_t cur_stream() const { return launch_stream ? launch_stream : stream; }
    // mvlm_set_stream: orders the context's work across a change of stream.  A context that has only ever run on the null
    // stream (which cannot be destroyed) records nothing; from its first change of stream on, every entry point leaves
    // switch_event behind the work it enqueued (MvlmOrderGuard), so a later change never touches the previous stream's
    // handle - its owner may have destroyed it by then.
    // surface snap: which upload the scratch's de-indexed triangle copy and vertex -> triangle table were made for
    unsigned long long snap_mesh_uid = 0;
    const void* snap_soup = nullptr;
    const void* snap_vert_tri = nullptr;
    hipEvent_t switch_event = nullptr;
    bool track_order = false;      // the context has been moved between streams: record switch_event at every exit
    bool order_recorded = false;   // switch_event stands behind everything enqueued so far
    std::mutex mu;
    std::string err;
    CnnState cnn;
    // grow-only internal scratch (raster bins, transformed vertices, small staging)
    std::map<std::string, std::pair<void*, size_t>> scratch;
    float* kparts_ws[2] = {nullptr, nullptr};      // split-K over workgroups: partial tiles, arrival counters
    unsigned* kparts_cnt[2] = {nullptr, nullptr};  // ([0] main launch stream, [1] the executor's side stream)
    int conv_force_variant = -1;            // >= 0: mvlm_conv_bench times exactly this kernel variant
    std::vector<ConvOverride> conv_overrides;  // tools/tune_in_network.py: kernel variant per (shape, kind), before any table
    unsigned long long conv_attr_mask = 0;  // conv variants whose launch attributes are set on this ctx's device
    size_t raster_keys_clean_cap = 0;  // capacity of the "raster.keys" scratch when it was last left all-EMPTY (0: not known clean)
    int render_subpixel_bits = 8;  // vertex snap 2^-bits pixel (GL_SUBPIXEL_BITS of the OpenGL to match)
    int render_shading = 0;  // 0: unlit nearest-texel RGB (reference), 1: build-defined geometry shading
    int* render_overflow_host = nullptr;  // pinned; written asynchronously by mvlm_render
    bool render_profiling = false;        // mvlm_render_set_profiling: HIP events around each render's kernels
    std::vector<RenderProfileRec> render_prof;
    std::vector<hipEvent_t> render_events;
    size_t render_event_cursor = 0;
    // device buffers of freed meshes, reused by the next upload: a folder of scans would otherwise pay
    // four hipMalloc + four (device-synchronising) hipFree per scan
    struct PoolEntry {
        void* p;
        size_t cap;
        hipEvent_t freed;  // recorded on the launch stream when the owning mesh was freed: work enqueued before may read p
    };
    std::vector<PoolEntry> mesh_pool;  // oldest first
    size_t mesh_pool_bytes = 0;
    // mesh uploads: pinned staging (two slots) -> hipMemcpyAsync on a stream of their own, so a reader thread's upload of
    // the next scan runs beside the current scan's kernels and never makes a host wait for the launch stream
    std::mutex upload_mu;  // staging slots + upload stream; taken before `mu`, never the other way round
    hipStream_t upload_stream = nullptr;
    void* upload_stage[2] = {nullptr, nullptr};
    size_t upload_stage_cap[2] = {0, 0};
    hipEvent_t upload_stage_done[2] = {nullptr, nullptr};
    int upload_stage_next = 0;
    // JPEG textures decoded on the device (jpeg.hip): scratch of the upload path, guarded by upload_mu
    void* jpeg_scratch = nullptr;
    size_t jpeg_scratch_cap = 0;
    int32_t* jpeg_flags_host = nullptr;  // pinned
    std::vector<hipEvent_t> event_free;  // recycled (timing-disabled) events
    hipEvent_t take_event() {
        if (!event_free.empty()) {
            hipEvent_t e = event_free.back();
            event_free.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        return e;
    }
    // The message of a failed call belongs to the calling thread: reader threads upload meshes while another thread
    // launches (predict_files), and both may fail.  `err` keeps the context's latest message (under err_mu) for a
    // thread that has none of its own; mvlm_last_error hands out the calling thread's copy.
    std::mutex err_mu;
    int fail(const std::string& m);
    void* get_scratch(const char* name, size_t bytes);
};

inline int mvlm_ctx::fail(const std::string& m) {
    {
        std::lock_guard<std::mutex> lock(err_mu);
        err = m;
    }
    mvlm_thread_error() = m;
    mvlm_thread_error_ctx() = this;
    return 1;
}

// End of every entry point (declared by MVLM_ENTER, destroyed before the context's mutex is released): once a context has
// been moved between streams, leave switch_event behind whatever this call enqueued on the current one.
struct MvlmOrderGuard {
    mvlm_ctx* ctx;
    ~MvlmOrderGuard() {
        if (!ctx->track_order || !ctx->switch_event) return;
        if (hipEventRecord(ctx->switch_event, ctx->stream) == hipSuccess)
            ctx->order_recorded = true;
        else
            (void)hipGetLastError();
    }
};

// a mesh's device copy is complete once its `ready` event has fired: consumers make their stream wait for it
inline int mvlm_mesh_wait_ready(mvlm_ctx* ctx, const mvlm_mesh* m, hipStream_t stream) {
    if (m->ready && hipStreamWaitEvent(stream, m->ready, 0) != hipSuccess) return ctx->fail("mesh: hipStreamWaitEvent failed");
    return 0;
}

// jpeg.hip (the host parser and its declarations: jpeg_plan.h / jpeg_plan.cpp)
int mvlm_jpeg_run(mvlm_ctx* ctx, MvlmJpegPlan& plan, const uint8_t* stage_pinned, uint8_t* rgb_dev, hipStream_t stream,
                  std::string& why, int* rounds_out);

// conv_mfma.hip
constexpr long MVLM_KPARTS_MAX_TILES = 2048;  // output tiles of a launch that divides K over workgroups
constexpr long MVLM_KPARTS_MAX_PARTS = 4096;  // tiles x parts (4 KB of partial sums each)
int mvlm_conv_kparts_workspace(mvlm_ctx* ctx, float** ws, unsigned** cnt);
int mvlm_launch_conv(mvlm_ctx* ctx, const ConvArgs& a, int* variant_out);
bool mvlm_conv_in2_ok(const mvlm_ctx* ctx, const ConvArgs& a);  // this launch would run on the tile that can add ConvArgs::in2 on its load
bool mvlm_conv_can_pool(const mvlm_ctx* ctx, const ConvArgs& a);  // the variant this launch would use can also emit the 2x2 max-pooled tensor
int mvlm_conv_kind(const ConvArgs& a);
bool mvlm_conv_variant_can_pool(int variant);
// two independent convolutions in one grid (conv_kernel.h: conv_pair_kernel)
constexpr int MVLM_CONV_PAIR_FLAG = 0x1000;  // variant code of a paired launch: flag | base id | lg(kparts0) << 8 | lg(kparts1) << 10
int mvlm_conv_pair_variant(const ConvArgs& a0, const ConvArgs& a1, int mode);
int mvlm_launch_conv_pair(mvlm_ctx* ctx, const ConvArgs& a0, const ConvArgs& a1, int pair_variant);
int mvlm_conv_amax_parts(int H, int W);  // partials per (image, channel) the argmax epilogue writes
const char* mvlm_conv_variant_name_impl(int v);

// conv_fast.hip (opt-in reduced-cost arithmetic)
// channel paddings of the opt-in bf16x3 kernel (its split weights are packed separately from the exact ones): 16-channel
// k-chunks, 64-channel tiles.  Served: 16..256 input channels, output padding waste of at most 1.6x (84 -> 128 yes, 32 -> 64 no).
inline int mvlm_fast_cin_pad(int cin) { return (cin + 15) / 16 * 16; }
inline int mvlm_fast_cout_pad(int cout) { return cout <= 32 ? 32 : (cout + 63) / 64 * 64; }  // (32: the f16x2 form's 32-channel tile)
inline bool mvlm_fast_channels_ok(int cin, int cout) {
    return cin >= 16 && cin <= 256 && cout > 0 && cout * 8 >= mvlm_fast_cout_pad(cout) * 5;
}
bool mvlm_conv_fast_ok(const ConvArgs& a, int splits);  // splits: 3 = bf16x3 ("fast"), 2 = f16x2 ("fast16")
int mvlm_launch_conv_fast(mvlm_ctx* ctx, const ConvArgs& a, const unsigned short* wq_dev, int splits = 3, float unscale = 1.f);
constexpr int MVLM_CONV_VARIANT_FAST = 62;    // id reported for launches of the bf16x3 kernel
constexpr int MVLM_CONV_VARIANT_FAST16 = 61;  // ... of the f16x2 kernel

// small kernels (misc.hip)
int mvlm_launch_pack_input(mvlm_ctx* ctx, const float* images, int n, const int* sel4, int c, float* out);
int mvlm_launch_maxpool2(mvlm_ctx* ctx, const float* in, int planes, int H, int W, float* out);
int mvlm_launch_amax_final(mvlm_ctx* ctx, const float* val, const int* idx, int n_img, int view0, int n_views_total,
                           int nl, int parts, int size, float* maxima, int* best_idx = nullptr);
// "moment" refinement of the fused argmax's peaks from conv10's output (misc.hip: moment_refine_kernel)
// f16x2 precision: where *flag != 0 the pass's maxima scores (or the first heatmap value of every plane) become NaN
int mvlm_launch_fp16_poison(mvlm_ctx* ctx, const unsigned* flag, float* maxima, int n_img, int view0, int n_views_total, int nl,
                            float* heat);
int mvlm_launch_moment_refine(mvlm_ctx* ctx, const float* x10, int n_img, int nl, int cin_pad, int cout_pad, const float* const w_par[4],
                              const float* bias, const int* best_idx, int view0, int n_views_total, float* maxima);
