// Graph executor for the landmark-heatmap network: walks the fixed two-stack hourglass
// architecture of the reference (MVLMModel.forward, src/mvlm/prediction/paulsenpredictor.py
// :404-432; ResidualBlock :267-273; HourGlassModule :301-361) and issues one fused
// convolution launch per conv layer (138 live convs; conv8 is dead at inference, :204-205).
//
// What is fused where (nothing else touches HBM):
//   BatchNorm+ReLU of a pre-activation block  -> consumer conv's input staging
//   bias / post-BN+ReLU (conv1, conv5, conv9)  -> producer conv's epilogue
//   torch.cat(out1,out2,out3) + residual       -> channel-slice writes + residual add in the
//                                                 three convs' epilogues
//   F.interpolate(x2, nearest) + skip add      -> 2x2 scatter in the epilogues of the last
//                                                 block of the lower hourglass level
//   F.interpolate before conv11                -> conv11 reads its input through >>1
//   np.argmax per (view, landmark)             -> conv11's epilogue + a tiny reduction
//   F.max_pool2d(x, 2, 2)                       -> 2x2 max in the producer's epilogue (tiles of 32-pixel
//                                                 rows; the 4-row tiles of small batches use the pool kernel)
//
// All views of a batch go through each layer together (pixels of all views form the GEMM's
// N dimension), so even the 4x4 hourglass level fills MFMA tiles.
#include <cstdlib>

#include "common.h"

namespace {

// the hourglass runs its lower pyramid beside the skip blocks when a 32x32 level holds at most this many
// pixels over the whole batch (= 32 views)
constexpr long CONCURRENT_MAX_PIXELS_32 = 32l * 32 * 32;

struct Tensor {
    float* p = nullptr;
    int C = 0, S = 0;
    size_t off = 0, elems = 0;
};

struct Block {
    size_t off, size;
    bool free;
    int arena;      // 0: main stream, 1: side stream - a block is only ever reused by kernels of its own arena's stream
    bool deferred;  // released while the streams ran apart: becomes free at the join
};

struct Exec {
    mvlm_ctx* ctx;
    CnnState& st;
    int B;
    char* ws;
    size_t ws_bytes;
    bool dry;
    size_t cursor = 0, high = 0;
    std::vector<Block> blocks;
    int rc = 0;
    int arena = 0;        // arena new tensors come from = stream the next launches go to
    bool forked = false;
    // MVLM_POOL_KERNEL_ONLY=1 (tests; launch-by-launch passes only - a captured graph keeps what it was captured with): every
    // max_pool2d as its own launch behind the block, the form the fused epilogues are compared with bit for bit
    const bool no_pool_fusion = [] { const char* e = getenv("MVLM_POOL_KERNEL_ONLY"); return e && e[0] == '1'; }();
    const bool scatter_only = [] { const char* e = getenv("MVLM_SCATTER_ONLY"); return e && e[0] == '1'; }();
    const int consumer_add_max_batch = [] { const char* e = getenv("MVLM_CONSUMER_ADD_MAX_BATCH"); return e ? atoi(e) : 16; }();

    Exec(mvlm_ctx* c, int batch, void* w, size_t wb, bool d) : ctx(c), st(c->cnn), B(batch), ws((char*)w), ws_bytes(wb), dry(d) {}

    Tensor alloc_raw(size_t bytes, int C, int S) {
        bytes = (bytes + 255) / 256 * 256;
        Tensor t;
        t.C = C;
        t.S = S;
        t.elems = bytes / 4;
        for (auto& b : blocks)
            if (b.free && b.size == bytes && b.arena == arena) {
                b.free = false;
                t.off = b.off;
                t.p = dry ? nullptr : reinterpret_cast<float*>(ws + b.off);
                return t;
            }
        blocks.push_back({cursor, bytes, false, arena, false});
        t.off = cursor;
        cursor += bytes;
        if (cursor > high) high = cursor;
        if (!dry && cursor > ws_bytes) {
            rc = ctx->fail("cnn: workspace too small (mvlm_cnn_workspace_bytes)");
            t.p = nullptr;
            return t;
        }
        t.p = dry ? nullptr : reinterpret_cast<float*>(ws + t.off);
        return t;
    }
    Tensor alloc(int C, int S) { return alloc_raw(size_t(B) * C * S * S * 4, C, S); }
    // Stream order makes "release, then reuse" safe within one stream.  While the two streams run apart
    // a main-arena tensor may still be read by kernels of the other stream (the branch's inputs), so
    // main-arena releases wait for the join; side-arena tensors are only touched by the side stream
    // until the join and can be recycled there at once.
    void release(const Tensor& t) {
        for (auto& b : blocks)
            if (b.off == t.off && !b.free && !b.deferred) {
                if (forked && b.arena == 0)
                    b.deferred = true;
                else
                    b.free = true;
                return;
            }
    }

    // ---- two-stream execution: fork() .. side() .. main() .. join() ---------------------------
    hipEvent_t next_event() {
        if (st.sync_cursor >= st.sync_events.size()) st.sync_events.resize(st.sync_cursor + 1, nullptr);
        hipEvent_t& e = st.sync_events[st.sync_cursor++];
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            rc = ctx->fail("cnn: hipEventCreate failed");
            return nullptr;
        }
        return e;
    }
    // everything enqueued on the main stream so far happens before what the side stream gets from now on
    void fork() {
        forked = true;
        if (dry || rc) return;
        if (!st.side_stream) {
            // highest priority: the side stream carries the chain of short launches; whenever a workgroup slot
            // frees up under the main stream's big launches, the chain's workgroups should get it first
            int lo_prio = 0, hi_prio = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
            if (hipStreamCreateWithPriority(&st.side_stream, hipStreamNonBlocking, hi_prio) != hipSuccess) {
                rc = ctx->fail("cnn: hipStreamCreate failed");
                return;
            }
        }
        hipEvent_t e = next_event();
        if (rc) return;
        if (hipEventRecord(e, ctx->stream) != hipSuccess || hipStreamWaitEvent(st.side_stream, e, 0) != hipSuccess)
            rc = ctx->fail("cnn: fork of the side stream failed");
    }
    void side() {
        arena = 1;
        if (!dry) ctx->launch_stream = st.side_stream;
    }
    void main() {
        arena = 0;
        ctx->launch_stream = nullptr;
    }
    // the main stream continues after everything the side stream was given
    void join() {
        main();
        forked = false;
        for (auto& b : blocks)
            if (b.deferred) {
                b.deferred = false;
                b.free = true;
            }
        if (dry || rc) return;
        hipEvent_t e = next_event();
        if (rc) return;
        if (hipEventRecord(e, st.side_stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, e, 0) != hipSuccess)
            rc = ctx->fail("cnn: join of the side stream failed");
    }

    const int32_t* d(int slot) const { return &st.desc[size_t(slot) * MVLM_CONV_DESC_INTS]; }
    const float* blob(int off) const { return off < 0 ? nullptr : st.blob + off; }

    // weights / BN / bias / shape of `slot` into `a`
    int fill(int slot, const Tensor& x, ConvArgs& a, int S) {
        if (rc) return rc;
        const int32_t* r = d(slot);
        if (!r[0]) return rc = ctx->fail("cnn: conv slot not present in the packed weights");
        a.in = x.p;
        a.in_ctot = x.C;
        a.in_coff = 0;
        a.cin = r[1];
        a.cout = r[2];
        a.ksize = r[3];
        a.cin_pad = r[4];
        a.cout_pad = r[5];
        a.w = blob(r[6]);
        a.bias = blob(r[7]);
        a.pre_scale = blob(r[8]);
        a.pre_shift = blob(r[9]);
        a.post_scale = blob(r[10]);
        a.post_shift = blob(r[11]);
        a.B = B;
        a.H = a.W = S;
        if (x.C != a.cin) return rc = ctx->fail("cnn: channel mismatch between graph and packed weights");
        return 0;
    }
    bool runs_fast(int slot, const ConvArgs& a) const {
        const std::vector<long long>& off = st.fast == 2 ? st.fast16_off : st.fast_off;
        return st.fast && size_t(slot) < off.size() && off[size_t(slot)] >= 0 && mvlm_conv_fast_ok(a, st.fast == 2 ? 2 : 3);
    }
    // per-launch HIP events (profiling runs only)
    int prof_begin(hipEvent_t& e0, hipEvent_t& e1) {
        if (!st.profiling) return 0;
        if (st.event_cursor + 2 > st.event_pool.size()) st.event_pool.resize(st.event_cursor + 2, nullptr);
        for (int k = 0; k < 2; ++k) {
            hipEvent_t& e = st.event_pool[st.event_cursor + k];
            if (!e && hipEventCreate(&e) != hipSuccess) return rc = ctx->fail("cnn: hipEventCreate failed");
        }
        e0 = st.event_pool[st.event_cursor];
        e1 = st.event_pool[st.event_cursor + 1];
        st.event_cursor += 2;
        hipEventRecord(e0, ctx->cur_stream());
        return 0;
    }
    static double conv_flops(const ConvArgs& a) { return 2.0 * a.cin * a.cout * a.ksize * a.ksize * double(a.H) * a.W * a.B * a.n_par; }

    // fill weights / BN / bias of `slot` into `a` and launch
    int conv(int slot, const Tensor& x, ConvArgs a, int S) {
        if (fill(slot, x, a, S)) return rc;
        if (dry) return 0;
        int variant = -1;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof_begin(e0, e1)) return rc;
        if (runs_fast(slot, a)) {
            if (st.fast == 2) {
                variant = MVLM_CONV_VARIANT_FAST16;
                if (mvlm_launch_conv_fast(ctx, a, st.fast16_blob + st.fast16_off[size_t(slot)], 2, st.fast16_unscale[size_t(slot)])) return rc = 1;
            } else {
                variant = MVLM_CONV_VARIANT_FAST;
                if (mvlm_launch_conv_fast(ctx, a, st.fast_blob + st.fast_off[size_t(slot)])) return rc = 1;
            }
        } else if (mvlm_launch_conv(ctx, a, &variant)) {
            return rc = 1;
        }
        if (st.profiling) {
            hipEventRecord(e1, ctx->cur_stream());
            ConvProfileRec rec{slot, variant, conv_flops(a), e0, e1};
            const short sh[6] = {short(a.ksize), short(a.cin_pad), short(a.cout_pad), short(a.H), short(mvlm_conv_kind(a)), short(B > 32767 ? 32767 : B)};
            for (int k = 0; k < 6; ++k) rec.shape[k] = sh[k];
            st.prof.push_back(rec);
        }
        return 0;
    }

    // the shared launch of two independent convolutions (conv_mfma.hip: mvlm_conv_pair_variant), or -1
    int pair_variant(int slot0, const Tensor& x0, ConvArgs a0, int S0, int slot1, const Tensor& x1, ConvArgs a1, int S1) {
        if (st.pairing == 0 || fill(slot0, x0, a0, S0) || fill(slot1, x1, a1, S1)) return -1;
        if (runs_fast(slot0, a0) || runs_fast(slot1, a1)) return -1;
        return mvlm_conv_pair_variant(a0, a1, st.pairing);
    }

    // two independent convolutions: one grid when `pv` names a shared kernel variant, else one launch each
    int conv_pair(int pv, int slot0, const Tensor& x0, ConvArgs a0, int S0, int slot1, const Tensor& x1, ConvArgs a1, int S1) {
        if (pv < 0) {
            conv(slot0, x0, a0, S0);
            return conv(slot1, x1, a1, S1);
        }
        if (fill(slot0, x0, a0, S0) || fill(slot1, x1, a1, S1)) return rc;
        if (dry) return 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof_begin(e0, e1)) return rc;
        if (mvlm_launch_conv_pair(ctx, a0, a1, pv)) return rc = 1;
        if (st.profiling) {
            hipEventRecord(e1, ctx->cur_stream());
            st.prof.push_back({slot0, pv, conv_flops(a0) + conv_flops(a1), e0, e1});  // one record: both problems' work, the first one's slot
        }
        return 0;
    }

    // One pre-activation residual block (paulsenpredictor.py:267-273), planned (tensors allocated, launch arguments of
    // its convolutions filled in) and then issued - alone (rb) or conv by conv beside an independent block (rb_pair).
    // hi == nullptr : y [cout@S]
    // hi != nullptr : writes upsample2x(y) + *hi into *hi in place ([cout@2S], the hourglass
    //                 skip tensor), y stays empty
    // pooled != nullptr : additionally max_pool2d(y, 2, 2) in *pooled - fused into the three
    //                 convs' epilogues when their kernel variants support it, else by the pool
    //                 kernel; with keep_full == false the full-resolution y is then never written
    //                 (it is still allocated: a resampling block keeps its 1x1 output there)
    struct RbPlan {
        int base = 0, S = 0, cout = 0;
        bool resample = false;
        Tensor x, y, t1, t2;
        Tensor* pooled = nullptr;
        ConvArgs ar, a[3];   // the 1x1 resample (when present), conv1..conv3
        bool want_pool = false, keep_full = true;
    };
    RbPlan plan_rb(int rb_index, const Tensor& x, Tensor* hi, Tensor* pooled = nullptr, bool keep_full = true) {
        RbPlan P;
        P.base = 1 + 4 * rb_index;
        P.S = x.S;
        P.x = x;
        P.cout = d(P.base + 1)[2] * 2;
        const int cout = P.cout, h = cout / 2, q = cout / 4;
        P.resample = d(P.base)[0] != 0;
        if (!hi) P.y = alloc(cout, P.S);
        if (P.resample && hi) {
            rc = ctx->fail("cnn: resampling block cannot scatter");
            return P;
        }
        P.t1 = alloc(h, P.S);
        P.t2 = alloc(q, P.S);
        if (P.resample) {
            P.ar.out = P.y.p;
            P.ar.out_ctot = cout;
        }
        const Tensor& res = P.resample ? P.y : P.x;
        const int coffs[3] = {0, h, h + q};
        for (int j = 0; j < 3; ++j) {
            ConvArgs& a = P.a[j];
            a.res1 = res.p;
            a.res1_ctot = res.C;
            a.res1_coff = coffs[j];
            if (hi) {
                a.out = hi->p;
                a.out_ctot = hi->C;
                a.out_coff = coffs[j];
                a.up_out = 1;
                a.skip = hi->p;
                a.skip_ctot = hi->C;
                a.skip_coff = coffs[j];
            } else {
                a.out = P.y.p;
                a.out_ctot = cout;
                a.out_coff = coffs[j];
            }
        }
        P.a[0].out_raw = P.t1.p;
        P.a[0].raw_ctot = h;
        P.a[1].out_raw = P.t2.p;
        P.a[1].raw_ctot = q;
        P.want_pool = pooled && !hi;
        if (P.want_pool)
            for (ConvArgs& a : P.a) a.pool_hint = 1;  // the variant choice knows the pooled tensor is wanted, fused or not
        P.keep_full = keep_full;
        P.pooled = pooled;
        if (P.want_pool) *pooled = alloc(cout, P.S / 2);
        return P;
    }
    const Tensor& rb_input(const RbPlan& P, int j) const { return j == 0 ? P.x : (j == 1 ? P.t1 : P.t2); }
    // the pooled tensor comes out of the three convolutions' epilogues (can_fuse: every one of their kernel variants can) ...
    void set_pool(RbPlan& P, bool can_fuse) {
        if (!P.want_pool || !can_fuse) return;
        for (ConvArgs& a : P.a) {
            a.pool_out = P.pooled->p;
            a.pool_ctot = P.cout;
            a.pool_coff = a.out_coff;
            if (!P.keep_full) a.out = nullptr;
        }
    }
    // ... or out of the pool kernel behind the block
    void finish_rb(RbPlan& P) {
        release(P.t1);
        release(P.t2);
        if (P.want_pool && !P.a[0].pool_out && !dry && !rc) pool_kernel(P.y.p, P.cout, P.S, P.pooled->p);
    }
    // F.max_pool2d as its own launch (the block's kernel variants cannot pool in their epilogues); profiled as slot -1
    void pool_kernel(const float* in, int channels, int S, float* out) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof_begin(e0, e1)) return;
        if (mvlm_launch_maxpool2(ctx, in, B * channels, S, S, out)) {
            rc = 1;
            return;
        }
        if (st.profiling) {
            hipEventRecord(e1, ctx->cur_stream());
            ConvProfileRec rec{-1, -1, 0.0, e0, e1};
            rec.shape[2] = short(channels);
            rec.shape[3] = short(S);
            rec.shape[5] = short(B > 32767 ? 32767 : B);
            st.prof.push_back(rec);
        }
    }

    Tensor rb(int rb_index, const Tensor& x, Tensor* hi, Tensor* pooled = nullptr, bool keep_full = true) {
        RbPlan P = plan_rb(rb_index, x, hi, pooled, keep_full);
        if (rc) return P.y;
        if (P.resample) conv(P.base, x, P.ar, P.S);
        bool fuse = P.want_pool;
        for (int j = 0; j < 3 && fuse; ++j) fuse = pool_fusable(P.base + 1 + j, rb_input(P, j), P.a[j], P.S);
        set_pool(P, fuse);
        for (int j = 0; j < 3; ++j) conv(P.base + 1 + j, rb_input(P, j), P.a[j], P.S);
        finish_rb(P);
        return P.y;
    }

    // Two residual blocks that do not depend on each other (a hourglass level's skip block and the first block of the
    // next lower level, paulsenpredictor.py:301-361): conv j of both in ONE launch wherever a shared kernel variant is known
    // to be faster than two launches (st.pairing; the measured table conv_pair_tuned.h), else conv by conv as usual.
    // yA = rb(rbA, xA), yB = rb(rbB, xB) with its pooled copy in *pooledB.
    void rb_pair(int rbA, const Tensor& xA, Tensor& yA, int rbB, const Tensor& xB, Tensor& yB, Tensor* pooledB) {
        RbPlan PA = plan_rb(rbA, xA, nullptr), PB = plan_rb(rbB, xB, nullptr, pooledB);
        yA = PA.y;
        yB = PB.y;
        if (rc) return;
        if (PA.resample) conv(PA.base, xA, PA.ar, PA.S);
        if (PB.resample) conv(PB.base, xB, PB.ar, PB.S);
        int pv[3];
        bool fuse = PB.want_pool;
        for (int j = 0; j < 3; ++j) {
            pv[j] = pair_variant(PA.base + 1 + j, rb_input(PA, j), PA.a[j], PA.S, PB.base + 1 + j, rb_input(PB, j), PB.a[j], PB.S);
            if (fuse) fuse = pv[j] >= 0 ? (!no_pool_fusion && mvlm_conv_variant_can_pool(pv[j])) : pool_fusable(PB.base + 1 + j, rb_input(PB, j), PB.a[j], PB.S);
        }
        set_pool(PB, fuse);
        for (int j = 0; j < 3; ++j)
            conv_pair(pv[j], PA.base + 1 + j, rb_input(PA, j), PA.a[j], PA.S, PB.base + 1 + j, rb_input(PB, j), PB.a[j], PB.S);
        finish_rb(PA);
        finish_rb(PB);
    }

    // the four parity slots of conv11 differ in their weights only (same shapes, one bias): they can share a launch
    bool parity_weights_uniform(int slot0) const {
        const int32_t* r0 = d(slot0);
        for (int q = 1; q < 4; ++q) {
            const int32_t* r = d(slot0 + q);
            if (!r[0] || r[1] != r0[1] || r[2] != r0[2] || r[3] != r0[3] || r[4] != r0[4] || r[5] != r0[5] || (r[7] < 0) != (r0[7] < 0)) return false;
            for (int k = 8; k <= 11; ++k)
                if (r[k] >= 0 || r0[k] >= 0) return false;  // plain conv + bias layers
        }
        return true;
    }

    // would conv `slot` on this input run a kernel variant that can also emit the pooled tensor?
    bool pool_fusable(int slot, const Tensor& x, ConvArgs a, int S) {
        if (no_pool_fusion) return false;
        const int32_t* r = d(slot);
        a.cin = r[1];
        a.cout = r[2];
        a.ksize = r[3];
        a.cin_pad = r[4];
        a.cout_pad = r[5];
        a.B = B;
        a.H = a.W = S;
        (void)x;
        if (runs_fast(slot, a)) return !(S & 1);  // the split-operand kernels pool in their epilogues too (round 5)
        return mvlm_conv_can_pool(ctx, a);
    }

    // can the layer that consumes a hourglass's output (conv5 / conv9: 256 -> 256 @ 128x128) add the top level's low3 on its
    // load?  Exact precision only (the split-operand kernels have no second input), where the dispatcher's measured choice for
    // that layer is the tile that has the form; MVLM_SCATTER_ONLY=1 (tests) keeps the producer-side scatter everywhere.
    // Measured (profiles/r05_consumer_side_add.txt): the block that stops scattering saves ~59 us per hourglass at 8 and at 12
    // views (its epilogue is exposed in a latency-bound launch) and 244 us at 96; the second load per staged element costs the
    // consumer 2.3 % (20 / 40 / 302 us at 8 / 12 / 96 views) - a gain up to about 16 views per device batch, a loss beyond.
    bool consumer_side_add(int slot) {
        if (st.fast || scatter_only || B > consumer_add_max_batch) return false;
        ConvArgs a;
        Tensor x;
        x.C = d(slot)[1];
        x.S = 128;
        const int saved = rc;
        if (fill(slot, x, a, 128)) {
            rc = saved;
            return false;
        }
        a.out = reinterpret_cast<float*>(ws);  // (any non-null pointer: the question is about shapes)
        a.out_ctot = a.cout;
        return mvlm_conv_in2_ok(ctx, a);
    }

    // HourGlassModule.forward (paulsenpredictor.py:301-361).  rb0 = index of this
    // hourglass's rb1 in the canonical residual-block order.  Consumes nothing; the caller
    // releases x.
    // x_pooled = max_pool2d(x) (the producer of x emits it from its epilogue where it can).
    // low_out != nullptr (round 5): the level's last block on the way up (rb20) writes its plain output [256 @ 64x64] to
    // *low_out instead of scattering every value into its 2x2 block of up1; the layer that reads the hourglass's output
    // (conv5 / conv9) then adds it on its load (ConvArgs::in2) - the same sums, bit for bit, without the producer rewriting
    // four times its tile.
    Tensor hourglass(int rb0, const Tensor& x, const Tensor& x_pooled, Tensor* low_out = nullptr) {
        auto R = [&](int i) { return rb0 + i - 1; };
        // Small batches: the 32x32 .. 4x4 levels are a chain of ~60 short, latency-bound launches on few
        // workgroups.  They only depend on low1's pooled copy, and the 128x128 / 64x64 skip blocks (up1, up11)
        // do not depend on them, so the chain CAN run on the side stream beside those big launches
        // (mvlm_cnn_set_execution, concurrency = 1; off by default: measured on the MI355X the chain's workgroups
        // either wait for a CU the big launch has filled or, where they fit beside it, crawl behind its matrix
        // work - DESIGN.md 4.1).  Large batches fill the chip with every launch; there the order is sequential.
        const bool small_batch = long(B) * 32 * 32 <= CONCURRENT_MAX_PIXELS_32;
        const bool concurrent = st.concurrency != 0 && small_batch;
        Tensor lowt11, lowt12, lowt13, lowt14;
        Tensor up1, up11, up12, up13, up14, low1, low11, low12, low13, low14;
        if (!concurrent) {
            // way down: at every level the skip block (up1*) and the first block of the next lower level (low1*) read the
            // same tensor / its pooled copy and do not depend on each other - issued as pairs (rb_pair)
            rb_pair(R(1), x, up1, R(2), x_pooled, low1, &lowt11);
            rb_pair(R(3), low1, up11, R(4), lowt11, low11, &lowt12);
            release(low1);
            release(lowt11);
            rb_pair(R(5), low11, up12, R(6), lowt12, low12, &lowt13);
            release(low11);
            release(lowt12);
            rb_pair(R(7), low12, up13, R(8), lowt13, low13, &lowt14);
            release(low12);
            release(lowt13);
            rb_pair(R(9), low13, up14, R(10), lowt14, low14, nullptr);
            release(low13);
            release(lowt14);
        } else {
            low1 = rb(R(2), x_pooled, nullptr, &lowt11);
            fork();
            side();
            low11 = rb(R(4), lowt11, nullptr, &lowt12);
            release(lowt11);
            up12 = rb(R(5), low11, nullptr);
            release(low11);
            low12 = rb(R(6), lowt12, nullptr, &lowt13);
            release(lowt12);
            up13 = rb(R(7), low12, nullptr);
            release(low12);
            low13 = rb(R(8), lowt13, nullptr, &lowt14);
            release(lowt13);
            up14 = rb(R(9), low13, nullptr);
            release(low13);
            low14 = rb(R(10), lowt14, nullptr);
            release(lowt14);
        }
        Tensor low2 = rb(R(11), low14, nullptr);
        release(low14);
        rb(R(12), low2, &up14);  // add1 = up(low3) + up14, in place in up14
        release(low2);
        Tensor low21 = rb(R(13), up14, nullptr);
        release(up14);
        rb(R(14), low21, &up13);  // add2
        release(low21);
        Tensor low22 = rb(R(15), up13, nullptr);
        release(up13);
        rb(R(16), low22, &up12);  // add3
        release(low22);
        Tensor low23 = rb(R(17), up12, nullptr);
        release(up12);
        if (concurrent) {
            main();
            up11 = rb(R(3), low1, nullptr);
            release(low1);
            up1 = rb(R(1), x, nullptr);
            join();
        }
        rb(R(18), low23, &up11);  // add4
        release(low23);
        Tensor low24 = rb(R(19), up11, nullptr);
        release(up11);
        if (low_out)
            *low_out = rb(R(20), low24, nullptr);  // low3 of the top level; add5 happens on the consumer's load
        else
            rb(R(20), low24, &up1);  // add5
        release(low24);
        return up1;
    }

    // images: [B,256,256,4]; either maxima (+view0/n_total) or heat ([B,NL,256,256]) is set
    int forward(const float* images, const int* sel4, float* maxima, int view0, int n_total, float* heat) {
        const int NL = st.n_landmarks, C = st.in_channels;
        const int SLOT_CONV5 = 1 + 4 * 43;
        Tensor x0 = alloc(C, 256);
        if (!dry && mvlm_launch_pack_input(ctx, images, B, sel4, C, x0.p)) return 1;
        Tensor a0 = alloc(64, 256);
        {
            ConvArgs a;
            a.out = a0.p;
            a.out_ctot = 64;
            conv(0, x0, a, 256);  // conv1 + bn1 + relu
        }
        release(x0);
        Tensor p1;
        Tensor a1 = rb(0, a0, nullptr, &p1, /*keep_full=*/false);  // conv2 + max-pool; only the pooled tensor is consumed
        release(a0);
        release(a1);
        Tensor a2 = rb(1, p1, nullptr);  // conv3
        release(p1);
        Tensor r3p;
        Tensor r3 = rb(2, a2, nullptr, &r3p);  // conv4 (+ its pooled copy for the hourglass)
        release(a2);
        Tensor h1_low;
        const bool cs1 = consumer_side_add(SLOT_CONV5);
        Tensor h1 = hourglass(3, r3, r3p, cs1 ? &h1_low : nullptr);
        release(r3p);
        Tensor ll1 = alloc(256, 128);
        {
            ConvArgs a;
            a.out = ll1.p;
            a.out_ctot = 256;
            if (cs1) {
                a.in2 = h1_low.p;
                a.in2_ctot = h1_low.C;
            }
            conv(SLOT_CONV5, h1, a, 128);  // conv5 + bn2 + relu (on up1 + upsample(low3) when the add is the consumer's)
        }
        release(h1);
        if (cs1) release(h1_low);
        Tensor x6 = alloc(NL, 128);
        {
            ConvArgs a;
            a.out = x6.p;
            a.out_ctot = NL;
            conv(SLOT_CONV5 + 1, ll1, a, 128);  // conv6
        }
        Tensor sum = alloc(256, 128), sump;
        {
            ConvArgs a;  // conv7; sum_temp = (r3 + ll1) + x  (:422)
            a.res1 = r3.p;
            a.res1_ctot = 256;
            a.res2 = ll1.p;
            a.res2_ctot = 256;
            a.out = sum.p;
            a.out_ctot = 256;
            sump = alloc(256, 64);
            a.pool_hint = 1;
            const bool fused = pool_fusable(SLOT_CONV5 + 2, x6, a, 128);
            if (fused) {
                a.pool_out = sump.p;
                a.pool_ctot = 256;
            }
            conv(SLOT_CONV5 + 2, x6, a, 128);
            if (!fused && !dry && !rc) pool_kernel(sum.p, 256, 128, sump.p);
        }
        release(x6);
        release(r3);
        release(ll1);
        Tensor h2_low;
        const bool cs2 = consumer_side_add(SLOT_CONV5 + 3);
        Tensor h2 = hourglass(23, sum, sump, cs2 ? &h2_low : nullptr);
        release(sump);
        release(sum);
        Tensor x9 = alloc(256, 128);
        {
            ConvArgs a;
            a.out = x9.p;
            a.out_ctot = 256;
            if (cs2) {
                a.in2 = h2_low.p;
                a.in2_ctot = h2_low.C;
            }
            conv(SLOT_CONV5 + 3, h2, a, 128);  // conv9 + bn3 + relu
        }
        release(h2);
        if (cs2) release(h2_low);
        Tensor x10 = alloc(NL, 128);
        {
            ConvArgs a;
            a.out = x10.p;
            a.out_ctot = NL;
            conv(SLOT_CONV5 + 4, x9, a, 128);  // conv10
        }
        release(x9);
        {
            // conv11 over the nearest-upsampled conv10 output (:428-429).  When the packed weights
            // carry the four parity slots (2x2 kernels with pre-summed taps) each output parity is
            // one launch over the 128x128 tensor: 16 instead of 36 tap evaluations per low-res pixel.
            const int SLOT_PARITY = SLOT_CONV5 + 6;
            const bool parity = d(SLOT_PARITY)[0] != 0 && (d(SLOT_PARITY)[5] == 96 || d(SLOT_PARITY)[5] == 80 || d(SLOT_PARITY)[5] == 84);
            const int parts1 = parity ? mvlm_conv_amax_parts(128, 128) : mvlm_conv_amax_parts(256, 256);
            const int parts = parity ? 4 * parts1 : parts1;
            Tensor av, ai, abest;
            const bool moment = !heat && st.selection == 1;
            if (moment && !parity) return rc = ctx->fail("cnn: the fused moment selection needs conv11 in its parity form");
            if (!heat) {
                av = alloc_raw(size_t(B) * NL * parts * 4, 0, 0);
                ai = alloc_raw(size_t(B) * NL * parts * 4, 0, 0);
                if (moment || dry) abest = alloc_raw(size_t(B) * NL * 4, 0, 0);  // (the workspace query covers both selections)
            }
            // the four parities share their input tiles: ONE launch with the parities of a tile on neighbouring workgroups
            // (ConvArgs::n_par); MVLM_CONV11_PARITY_LAUNCHES=4 keeps one launch per parity (experiments)
            static const bool split_launches = [] { const char* e = getenv("MVLM_CONV11_PARITY_LAUNCHES"); return e && e[0] == '4'; }();
            const bool one_launch = parity && !split_launches && parity_weights_uniform(SLOT_PARITY);
            for (int pl = 0; pl < (parity && !one_launch ? 4 : 1); ++pl) {
                ConvArgs a;
                Tensor xin = x10;
                if (parity) {
                    a.sub_y = pl >> 1;
                    a.sub_x = pl & 1;
                    a.up_out = 2;
                    if (one_launch) {
                        a.n_par = 4;
                        for (int q = 0; q < 4; ++q) a.w_par[q] = blob(d(SLOT_PARITY + q)[6]);
                        a.amax_par_stride = parts1;
                    }
                } else {
                    a.up_in = 1;
                    xin.S = 256;
                }
                if (heat) {
                    a.out = heat;
                    a.out_ctot = NL;
                } else {
                    a.amax_val = av.p;
                    a.amax_idx = reinterpret_cast<int*>(ai.p);
                    a.amax_parts = parts;
                    a.amax_part0 = pl * parts1;
                }
                conv(parity ? SLOT_PARITY + pl : SLOT_CONV5 + 5, xin, a, parity ? 128 : 256);
            }
            if (!heat) {
                if (!dry && !rc &&
                    mvlm_launch_amax_final(ctx, av.p, reinterpret_cast<int*>(ai.p), B, view0, n_total, NL, parts, 256,
                                           maxima, moment ? reinterpret_cast<int*>(abest.p) : nullptr))
                    rc = 1;
                if (moment && !dry && !rc) {
                    // paulsenpredictor.py:129-156 on the planes whose peak lies more than 15 pixels inside: the 31x31 window is
                    // recomputed from x10 with the parity kernels' own arithmetic - no [N,NL,256,256] tensor
                    const int32_t* r0 = d(SLOT_PARITY);
                    const float* wq[4];
                    for (int q = 0; q < 4; ++q) wq[q] = blob(d(SLOT_PARITY + q)[6]);
                    if (mvlm_launch_moment_refine(ctx, x10.p, B, NL, r0[4], r0[5], wq, blob(r0[7]), reinterpret_cast<int*>(abest.p), view0,
                                                  n_total, maxima))
                        rc = 1;
                }
                release(av);
                release(ai);
                if (moment || dry) release(abest);
            }
        }
        release(x10);
        if (st.fast == 2 && !dry && !rc && mvlm_launch_fp16_poison(ctx, st.fast16_flag, heat ? nullptr : maxima, B, view0, n_total, NL, heat)) rc = 1;
        return rc;
    }
};

// One batch of views through the network.  Eager: the ~160 launches are enqueued one by one (two streams
// for small batches).  Replay: the same sequence, captured once per (buffers, shapes) as a hipGraph with
// the side stream's work as a parallel branch, is launched with one call - at small batches the host
// otherwise spends as long enqueueing as the GPU spends computing.  A key is captured the second time it is
// seen (the first, eager pass also performs the one-time launch-attribute set-up, which must not happen
// inside a capture).  Profiling (per-launch events) always runs eagerly.
int run_batch(mvlm_ctx* ctx, const float* img, int nb, const int* sel4, float* maxima, int view0, int n_total,
              float* heat, void* ws, size_t ws_bytes) {
    CnnState& st = ctx->cnn;
    auto eager = [&]() {
        st.sync_cursor = 0;
        Exec ex(ctx, nb, ws, ws_bytes, false);
        const int rc = ex.forward(img, sel4, maxima, view0, n_total, heat);
        ctx->launch_stream = nullptr;
        ++st.eager_runs;
        return rc;
    };
    if (!st.graph_mode || st.profiling) return eager();
    const void* out = heat ? static_cast<const void*>(heat) : static_cast<const void*>(maxima);
    CnnGraphEntry* ent = nullptr;
    for (auto& g : st.graphs)
        if (g.images == img && g.out == out && g.ws == ws && g.ws_bytes == ws_bytes && g.nb == nb && g.view0 == view0 &&
            g.n_total == n_total && g.heat == (heat ? 1 : 0) && g.sel4[0] == sel4[0] && g.sel4[1] == sel4[1] &&
            g.sel4[2] == sel4[2] && g.sel4[3] == sel4[3]) {
            ent = &g;
            break;
        }
    if (!ent) {
        constexpr size_t MAX_GRAPHS = 16;
        if (st.graphs.size() >= MAX_GRAPHS) {  // drop the oldest entry
            if (st.graphs.front().exec) hipGraphExecDestroy(st.graphs.front().exec);
            st.graphs.erase(st.graphs.begin());
        }
        CnnGraphEntry g;
        g.images = img;
        g.out = out;
        g.ws = ws;
        g.ws_bytes = ws_bytes;
        g.nb = nb;
        g.view0 = view0;
        g.n_total = n_total;
        g.heat = heat ? 1 : 0;
        for (int k = 0; k < 4; ++k) g.sel4[k] = sel4[k];
        st.graphs.push_back(g);
        ent = &st.graphs.back();
    }
    ++ent->uses;
    if (ent->exec) {
        MVLM_CHECK_HIP(ctx, hipGraphLaunch(ent->exec, ctx->stream));
        ++st.graph_replays;
        return 0;
    }
    if (ent->uses < 2) return eager();
    // capture.  The caller's stream may be the legacy null stream (torch's default), which cannot be captured:
    // the pass is recorded on a stream of our own and the resulting graph is launched on the caller's.
    auto note = [&](const char* what, hipError_t e) {
        if (st.graph_failures++ == 0)
            fprintf(stderr, "mvlm_hip: %s failed (%s); this pass runs launch by launch\n", what, hipGetErrorString(e));
        (void)hipGetLastError();
    };
    if (!st.capture_stream) {
        const hipError_t e = hipStreamCreateWithFlags(&st.capture_stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            st.capture_stream = nullptr;
            note("hipStreamCreate (capture stream)", e);
            return eager();
        }
    }
    st.sync_cursor = 0;
    hipStream_t user_stream = ctx->stream;
    hipError_t e_begin = hipStreamBeginCapture(st.capture_stream, hipStreamCaptureModeThreadLocal);
    if (e_begin != hipSuccess) {
        note("hipStreamBeginCapture", e_begin);
        return eager();
    }
    int rc;
    ctx->stream = st.capture_stream;
    {
        Exec ex(ctx, nb, ws, ws_bytes, false);
        rc = ex.forward(img, sel4, maxima, view0, n_total, heat);
        ctx->launch_stream = nullptr;
    }
    ctx->stream = user_stream;
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(st.capture_stream, &graph);
    if (rc) {  // the executor's own error (message already set); nothing was enqueued
        if (graph) hipGraphDestroy(graph);
        return rc;
    }
    if (e_end != hipSuccess || !graph) {
        note("hipStreamEndCapture", e_end);
        if (graph) hipGraphDestroy(graph);
        ent->uses = -1000000;  // do not try this key again
        return eager();
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e_inst != hipSuccess || !exec) {
        note("hipGraphInstantiate", e_inst);
        ent->uses = -1000000;
        return eager();
    }
    ent->exec = exec;
    ++st.graph_captures;
    MVLM_CHECK_HIP(ctx, hipGraphLaunch(ent->exec, ctx->stream));
    ++st.graph_replays;
    return 0;
}

int run_cnn(mvlm_ctx* ctx, const float* images, int n_views, const int32_t* chan_sel, float* maxima, float* heat,
            void* ws, size_t ws_bytes, int batch) {
    CnnState& st = ctx->cnn;
    MVLM_REQUIRE(ctx, st.loaded, "cnn: mvlm_cnn_load has not been called");
    MVLM_REQUIRE(ctx, images && n_views > 0 && batch > 0 && ws, "cnn: bad arguments");
    MVLM_REQUIRE(ctx, maxima || heat, "cnn: no output requested");
    int sel4[4] = {0, 0, 0, 0};
    for (int k = 0; k < st.in_channels; ++k) {
        MVLM_REQUIRE(ctx, chan_sel[k] >= 0 && chan_sel[k] < 4, "cnn: channel selector out of range");
        sel4[k] = chan_sel[k];
    }
    st.prof.clear();
    st.event_cursor = 0;
    // f16x2 precision: the range flag of this call starts clear (all device batches of the call share it: an overflow in
    // one batch poisons that batch's and the later batches' maxima, and mvlm_cnn_fast16_overflowed reports it afterwards)
    if (st.fast == 2) MVLM_CHECK_HIP(ctx, hipMemsetAsync(st.fast16_flag, 0, sizeof(unsigned), ctx->stream));
    const size_t img_elems = size_t(MVLM_IMAGE_SIZE) * MVLM_IMAGE_SIZE;
    for (int v0 = 0; v0 < n_views; v0 += batch) {
        const int nb = (n_views - v0) < batch ? (n_views - v0) : batch;
        float* h = heat ? heat + size_t(v0) * st.n_landmarks * img_elems : nullptr;
        const float* img = images + size_t(v0) * img_elems * 4;
        if (run_batch(ctx, img, nb, sel4, maxima, v0, n_views, h, ws, ws_bytes)) return 1;
    }
    return 0;
}

}  // namespace

extern "C" int mvlm_cnn_load(mvlm_ctx* ctx, const float* blob_host, size_t n_floats, const int32_t* desc_host,
                             int n_slots, int n_landmarks, int in_channels) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, blob_host && desc_host && n_floats > 0, "cnn_load: null weights");
    MVLM_REQUIRE(ctx, n_slots == 1 + 4 * 43 + 6 + 4, "cnn_load: descriptor table must have 183 conv slots");
    MVLM_REQUIRE(ctx, n_landmarks > 0 && in_channels >= 1 && in_channels <= 4, "cnn_load: bad landmark / channel count");
    CnnState& st = ctx->cnn;
    MVLM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (st.blob) {
        MVLM_CHECK_HIP(ctx, hipFree(st.blob));
        st.blob = nullptr;
    }
    st.loaded = false;
    st.fast = 0;
    st.fast_off.clear();
    if (st.fast_blob) {
        (void)hipFree(st.fast_blob);
        st.fast_blob = nullptr;
    }
    st.fast16_off.clear();
    st.fast16_unscale.clear();
    if (st.fast16_blob) {
        (void)hipFree(st.fast16_blob);
        st.fast16_blob = nullptr;
    }
    for (auto& g : st.graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);  // captured launches point into the old weight blob
    st.graphs.clear();
    st.desc.assign(desc_host, desc_host + size_t(n_slots) * MVLM_CONV_DESC_INTS);
    // validate every offset before anything is launched with it
    for (int s = 0; s < n_slots; ++s) {
        const int32_t* r = &st.desc[size_t(s) * MVLM_CONV_DESC_INTS];
        if (!r[0]) continue;
        MVLM_REQUIRE(ctx, r[1] > 0 && r[2] > 0 && r[3] >= 1 && r[3] <= 3, "cnn_load: bad conv shape");
        // output channels: multiples of 16 (32-row tiles, 64 + 16-row strip), or exactly 84 = 64 + 16 + 4 rows
        MVLM_REQUIRE(ctx, r[4] % 4 == 0 && r[4] >= r[1] && (r[5] % 16 == 0 || (r[5] == 84 && r[2] == 84 && (r[3] == 3 || r[3] == 2))) && r[5] >= r[2],
                     "cnn_load: bad padding");
        const size_t wsz = size_t(r[3]) * r[3] * r[4] * r[5];
        MVLM_REQUIRE(ctx, r[6] >= 0 && size_t(r[6]) + wsz <= n_floats, "cnn_load: weight offset out of range");
        const int offs[5] = {r[7], r[8], r[9], r[10], r[11]};
        const int lens[5] = {r[5], r[4], r[4], r[5], r[5]};
        for (int k = 0; k < 5; ++k)
            MVLM_REQUIRE(ctx, offs[k] < 0 || size_t(offs[k]) + lens[k] <= n_floats, "cnn_load: vector offset out of range");
    }
    MVLM_REQUIRE(ctx, st.desc[1] == in_channels, "cnn_load: conv1 input channels != in_channels");
    MVLM_REQUIRE(ctx, st.desc[size_t(1 + 4 * 43 + 5) * MVLM_CONV_DESC_INTS + 2] == n_landmarks,
                 "cnn_load: conv11 output channels != n_landmarks");
    MVLM_CHECK_HIP(ctx, hipMalloc(&st.blob, n_floats * sizeof(float)));
    MVLM_CHECK_HIP(ctx, hipMemcpy(st.blob, blob_host, n_floats * sizeof(float), hipMemcpyHostToDevice));
    st.n_landmarks = n_landmarks;
    st.in_channels = in_channels;
    st.loaded = true;
    return 0;
}

extern "C" size_t mvlm_cnn_workspace_bytes(mvlm_ctx* ctx, int batch) {
    std::lock_guard<std::mutex> lk(ctx->mu);  // host-only planning: no device work, 0 = failure
    if (!ctx->cnn.loaded || batch <= 0) return 0;
    Exec ex(ctx, batch, nullptr, 0, true);
    const int sel4[4] = {0, 0, 0, 0};
    ex.forward(nullptr, sel4, reinterpret_cast<float*>(8), 0, batch, nullptr);
    return ex.high + 256;
}

extern "C" int mvlm_cnn_maxima(mvlm_ctx* ctx, const float* images_dev, int n_views, const int32_t* chan_sel_host,
                               float* maxima_dev, void* workspace_dev, size_t workspace_bytes, int batch) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, maxima_dev && chan_sel_host, "cnn_maxima: null output / selector");
    return run_cnn(ctx, images_dev, n_views, chan_sel_host, maxima_dev, nullptr, workspace_dev, workspace_bytes, batch);
}

extern "C" int mvlm_cnn_heatmaps(mvlm_ctx* ctx, const float* images_dev, int n_views, const int32_t* chan_sel_host,
                                 float* heat_dev, void* workspace_dev, size_t workspace_bytes, int batch) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, heat_dev && chan_sel_host, "cnn_heatmaps: null output / selector");
    return run_cnn(ctx, images_dev, n_views, chan_sel_host, nullptr, heat_dev, workspace_dev, workspace_bytes, batch);
}

extern "C" int mvlm_cnn_set_execution(mvlm_ctx* ctx, int graph_mode, int concurrency) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, (graph_mode == 0 || graph_mode == 1) && (concurrency == 0 || concurrency == 1),
                 "cnn_set_execution: flags must be 0 or 1");
    CnnState& st = ctx->cnn;
    if (concurrency != st.concurrency) {  // captured graphs encode the launch order
        for (auto& g : st.graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        st.graphs.clear();
    }
    st.graph_mode = graph_mode;
    st.concurrency = concurrency;
    return 0;
}

extern "C" int mvlm_cnn_set_selection(mvlm_ctx* ctx, int method) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, method == MVLM_MAXIMA_SIMPLE || method == MVLM_MAXIMA_MOMENT, "cnn_set_selection: 0 simple, 1 moment");
    CnnState& st = ctx->cnn;
    MVLM_REQUIRE(ctx, st.loaded, "cnn_set_selection: mvlm_cnn_load comes first");
    if (method == MVLM_MAXIMA_MOMENT) {
        const int slot = 1 + 4 * 43 + 6;
        const int32_t* r = &st.desc[size_t(slot) * MVLM_CONV_DESC_INTS];
        MVLM_REQUIRE(ctx, r[0] != 0 && (r[5] == 96 || r[5] == 80 || r[5] == 84) && r[3] == 2,
                     "cnn_set_selection: the fused moment selection needs conv11's parity slots in the packed weights");
    }
    if (method != st.selection) {  // captured graphs encode the launches
        for (auto& g : st.graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        st.graphs.clear();
    }
    st.selection = method;
    return 0;
}

extern "C" int mvlm_cnn_set_pairing(mvlm_ctx* ctx, int mode) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, mode >= 0 && mode <= 2, "cnn_set_pairing: 0 never, 1 the measured table (default), 2 wherever one kernel variant serves both");
    CnnState& st = ctx->cnn;
    if (mode != st.pairing) {  // captured graphs encode the launches
        for (auto& g : st.graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        st.graphs.clear();
    }
    st.pairing = mode;
    return 0;
}

extern "C" int mvlm_cnn_execution_stats(mvlm_ctx* ctx, int64_t* eager_runs, int64_t* graph_captures,
                                        int64_t* graph_replays, int64_t* graph_failures) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    const CnnState& st = ctx->cnn;
    if (eager_runs) *eager_runs = st.eager_runs;
    if (graph_captures) *graph_captures = st.graph_captures;
    if (graph_replays) *graph_replays = st.graph_replays;
    if (graph_failures) *graph_failures = st.graph_failures;
    return 0;
}

// ---- opt-in "fast" precision ------------------------------------------------------------------------------------------------
extern "C" int mvlm_cnn_load_fast(mvlm_ctx* ctx, const uint16_t* blob_host, size_t n_u16, const int64_t* slot_offsets, int n_slots) {
    MVLM_ENTER(ctx);
    CnnState& st = ctx->cnn;
    MVLM_REQUIRE(ctx, st.loaded, "cnn_load_fast: mvlm_cnn_load comes first");
    MVLM_REQUIRE(ctx, blob_host && slot_offsets && n_u16 > 0 && size_t(n_slots) * MVLM_CONV_DESC_INTS == st.desc.size(),
                 "cnn_load_fast: bad arguments");
    for (int s = 0; s < n_slots; ++s) {
        if (slot_offsets[s] < 0) continue;
        const int32_t* r = &st.desc[size_t(s) * MVLM_CONV_DESC_INTS];
        MVLM_REQUIRE(ctx, r[0] && r[3] == 3 && mvlm_fast_channels_ok(r[1], r[2]), "cnn_load_fast: slot is not a fast-eligible 3x3 layer");
        const size_t need = size_t(mvlm_fast_cin_pad(r[1]) / 16) * 9 * 2 * 3 * size_t(mvlm_fast_cout_pad(r[2])) * 8;
        MVLM_REQUIRE(ctx, slot_offsets[s] % 8 == 0 && size_t(slot_offsets[s]) + need <= n_u16, "cnn_load_fast: offset out of range");
    }
    for (auto& g : st.graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    st.graphs.clear();
    if (st.fast == 1) st.fast = 0;
    if (st.fast_blob) {
        MVLM_CHECK_HIP(ctx, hipFree(st.fast_blob));
        st.fast_blob = nullptr;
    }
    MVLM_CHECK_HIP(ctx, hipMalloc(&st.fast_blob, n_u16 * sizeof(uint16_t)));
    MVLM_CHECK_HIP(ctx, hipMemcpy(st.fast_blob, blob_host, n_u16 * sizeof(uint16_t), hipMemcpyHostToDevice));
    st.fast_off.assign(slot_offsets, slot_offsets + n_slots);
    return 0;
}

// the f16x2 form's weights (mvlm_pack_fast_weights16): one blob, per conv slot the u16 offset (-1: stays exact) and the inverse scale
extern "C" int mvlm_cnn_load_fast16(mvlm_ctx* ctx, const uint16_t* blob_host, size_t n_u16, const int64_t* slot_offsets,
                                    const float* slot_unscale, int n_slots) {
    MVLM_ENTER(ctx);
    CnnState& st = ctx->cnn;
    MVLM_REQUIRE(ctx, st.loaded, "cnn_load_fast16: mvlm_cnn_load comes first");
    MVLM_REQUIRE(ctx, blob_host && slot_offsets && slot_unscale && n_u16 > 0 && size_t(n_slots) * MVLM_CONV_DESC_INTS == st.desc.size(),
                 "cnn_load_fast16: bad arguments");
    for (int s = 0; s < n_slots; ++s) {
        if (slot_offsets[s] < 0) continue;
        const int32_t* r = &st.desc[size_t(s) * MVLM_CONV_DESC_INTS];
        MVLM_REQUIRE(ctx, r[0] && r[3] == 3 && mvlm_fast_channels_ok(r[1], r[2]), "cnn_load_fast16: slot is not a fast-eligible 3x3 layer");
        const size_t need = size_t(mvlm_fast_cin_pad(r[1]) / 16) * 9 * 2 * 2 * size_t(mvlm_fast_cout_pad(r[2])) * 8;
        MVLM_REQUIRE(ctx, slot_offsets[s] % 8 == 0 && size_t(slot_offsets[s]) + need <= n_u16, "cnn_load_fast16: offset out of range");
        MVLM_REQUIRE(ctx, slot_unscale[s] > 0.f && slot_unscale[s] < 3.0e38f, "cnn_load_fast16: bad inverse scale");
    }
    for (auto& g : st.graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    st.graphs.clear();
    if (st.fast == 2) st.fast = 0;
    if (st.fast16_blob) {
        MVLM_CHECK_HIP(ctx, hipFree(st.fast16_blob));
        st.fast16_blob = nullptr;
    }
    if (!st.fast16_flag) {
        MVLM_CHECK_HIP(ctx, hipMalloc(&st.fast16_flag, sizeof(unsigned)));
        MVLM_CHECK_HIP(ctx, hipMemset(st.fast16_flag, 0, sizeof(unsigned)));
    }
    MVLM_CHECK_HIP(ctx, hipMalloc(&st.fast16_blob, n_u16 * sizeof(uint16_t)));
    MVLM_CHECK_HIP(ctx, hipMemcpy(st.fast16_blob, blob_host, n_u16 * sizeof(uint16_t), hipMemcpyHostToDevice));
    st.fast16_off.assign(slot_offsets, slot_offsets + n_slots);
    st.fast16_unscale.assign(slot_unscale, slot_unscale + n_slots);
    return 0;
}

// did an f16x2 launch of the last mvlm_cnn_maxima / mvlm_cnn_heatmaps call meet an activation outside fp16's range?  Waits for
// the context's stream (callers ask after their own wait for the results, when that costs nothing) and copies one word back.
extern "C" int mvlm_cnn_fast16_overflowed(mvlm_ctx* ctx, int* overflowed) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, overflowed, "cnn_fast16_overflowed: null output");
    *overflowed = 0;
    if (!ctx->cnn.fast16_flag) return 0;
    unsigned v = 0;
    MVLM_CHECK_HIP(ctx, hipMemcpyAsync(&v, ctx->cnn.fast16_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *overflowed = v != 0u;
    return 0;
}

extern "C" int mvlm_cnn_set_precision(mvlm_ctx* ctx, int fast) {
    MVLM_ENTER(ctx);
    CnnState& st = ctx->cnn;
    MVLM_REQUIRE(ctx, fast >= 0 && fast <= 2, "cnn_set_precision: 0 (exact fp32), 1 (bf16x3 split) or 2 (f16x2 split on the eligible layers)");
    MVLM_REQUIRE(ctx, fast != 1 || st.fast_blob, "cnn_set_precision: mvlm_cnn_load_fast has not been called");
    MVLM_REQUIRE(ctx, fast != 2 || st.fast16_blob, "cnn_set_precision: mvlm_cnn_load_fast16 has not been called");
    if (fast != st.fast) {  // captured graphs encode the kernels
        for (auto& g : st.graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        st.graphs.clear();
    }
    st.fast = fast;
    return 0;
}

extern "C" int mvlm_cnn_set_profiling(mvlm_ctx* ctx, int enabled) {
    MVLM_ENTER(ctx);
    ctx->cnn.profiling = enabled != 0;
    return 0;
}

extern "C" int mvlm_cnn_get_profile(mvlm_ctx* ctx, int32_t* slot, int32_t* variant, double* flops, float* ms, int cap) {
    std::lock_guard<std::mutex> lk(ctx->mu);  // returns a record count, -1 on failure
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    CnnState& st = ctx->cnn;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    int n = 0;
    for (const auto& r : st.prof) {
        if (n >= cap) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
        slot[n] = r.slot;
        variant[n] = r.variant;
        flops[n] = r.flops;
        ms[n] = t;
        ++n;
    }
    return n;
}

// shapes of the records mvlm_cnn_get_profile returns, in the same order: {ksize, cin_pad, cout_pad, size, kind, batch} per record
extern "C" int mvlm_cnn_get_profile_shapes(mvlm_ctx* ctx, int32_t* shapes6, int cap) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    int n = 0;
    for (const auto& r : ctx->cnn.prof) {
        if (n >= cap) break;
        for (int k = 0; k < 6; ++k) shapes6[n * 6 + k] = r.shape[k];
        ++n;
    }
    return n;
}

extern "C" const char* mvlm_conv_variant_name(int variant) { return mvlm_conv_variant_name_impl(variant); }
