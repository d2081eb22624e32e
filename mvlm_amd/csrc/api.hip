// Context, mesh upload and the single-convolution test hook of the C ABI
// (include/mvlm_hip.h).
#include <atomic>
#include <cstring>

#include <memory>

#include "common.h"

void* mvlm_ctx::get_scratch(const char* name, size_t bytes) {
    auto& e = scratch[name];
    if (e.second >= bytes && e.first) return e.first;
    if (e.first) {
        hipStreamSynchronize(stream);
        hipFree(e.first);
        e = {nullptr, 0};
    }
    void* p = nullptr;
    const size_t want = bytes + bytes / 4 + 256;  // headroom so growing view counts do not realloc every call
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    e = {p, want};
    return p;
}

extern "C" const char* mvlm_build_arch(void) { return "gfx950"; }

extern "C" int mvlm_ctx_create(int device, mvlm_ctx** out) {
    if (!out) return 1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return 2;
    if (hipSetDevice(device) != hipSuccess) return 3;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 4;
    auto* ctx = new mvlm_ctx();
    ctx->device = device;
    // the code objects in this library exist for gfx950 only - refuse anything else loudly
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        fprintf(stderr, "mvlm_hip: device %d is %s, this library is built for gfx950 (MI355X) only\n", device,
                prop.gcnArchName);
        delete ctx;
        return 5;
    }
    *out = ctx;
    return 0;
}

extern "C" void mvlm_ctx_destroy(mvlm_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->scratch)
        if (kv.second.first) hipFree(kv.second.first);
    if (ctx->cnn.blob) hipFree(ctx->cnn.blob);
    if (ctx->cnn.fast_blob) hipFree(ctx->cnn.fast_blob);
    if (ctx->cnn.fast16_blob) hipFree(ctx->cnn.fast16_blob);
    if (ctx->cnn.fast16_flag) hipFree(ctx->cnn.fast16_flag);
    if (ctx->switch_event) hipEventDestroy(ctx->switch_event);
    for (int i = 0; i < 2; ++i) {
        if (ctx->kparts_ws[i]) hipFree(ctx->kparts_ws[i]);
        if (ctx->kparts_cnt[i]) hipFree(ctx->kparts_cnt[i]);
    }
    for (auto& e : ctx->mesh_pool) {
        hipFree(e.p);
        if (e.freed) hipEventDestroy(e.freed);
    }
    for (hipEvent_t e : ctx->event_free) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (ctx->upload_stage[i]) hipHostFree(ctx->upload_stage[i]);
        if (ctx->upload_stage_done[i]) hipEventDestroy(ctx->upload_stage_done[i]);
    }
    if (ctx->upload_stream) hipStreamDestroy(ctx->upload_stream);
    if (ctx->jpeg_scratch) hipFree(ctx->jpeg_scratch);
    if (ctx->jpeg_flags_host) hipHostFree(ctx->jpeg_flags_host);
    if (ctx->render_overflow_host) hipHostFree(ctx->render_overflow_host);
    for (auto e : ctx->cnn.event_pool)
        if (e) hipEventDestroy(e);
    for (auto e : ctx->render_events)
        if (e) hipEventDestroy(e);
    for (auto e : ctx->cnn.sync_events)
        if (e) hipEventDestroy(e);
    for (auto& g : ctx->cnn.graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    if (ctx->cnn.side_stream) hipStreamDestroy(ctx->cnn.side_stream);
    if (ctx->cnn.capture_stream) hipStreamDestroy(ctx->cnn.capture_stream);
    delete ctx;
}

// The calling thread's last failure on this context (valid until that thread's next failing call); a thread without
// one of its own gets a private copy of the context's latest message - never a pointer into a string another thread
// may be reassigning.
extern "C" const char* mvlm_last_error(mvlm_ctx* ctx) {
    if (!ctx) return "null context";
    if (mvlm_thread_error_ctx() != ctx) {
        std::lock_guard<std::mutex> lock(ctx->err_mu);
        mvlm_thread_error() = ctx->err;
        mvlm_thread_error_ctx() = ctx;
    }
    return mvlm_thread_error().c_str();
}

// A context owns grow-only scratch (render bins and keys, transformed vertices, fusion staging) that its kernels reuse from
// call to call: safe while the calls are ordered on one stream.  Callers that move a context to ANOTHER stream (two
// pipelines of one device under different torch streams share the process-wide context) get that order kept for them: the
// new stream waits for everything the context enqueued before.  The event it waits for was recorded at the END of the
// previous entry points (MvlmOrderGuard), never here on the previous stream's handle - its owner may have destroyed that
// stream since.  Only the very first change records here: until then the context has lived on the stream it was created
// with, the null stream, which is never destroyed.
extern "C" int mvlm_set_stream(mvlm_ctx* ctx, void* hip_stream) {
    MVLM_ENTER(ctx);
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next != ctx->stream) {
        if (!ctx->switch_event) MVLM_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->switch_event, hipEventDisableTiming));
        if (!ctx->track_order) {
            MVLM_CHECK_HIP(ctx, hipEventRecord(ctx->switch_event, ctx->stream));  // (ctx->stream is still the null stream)
            ctx->order_recorded = true;
            ctx->track_order = true;
        }
        if (ctx->order_recorded) MVLM_CHECK_HIP(ctx, hipStreamWaitEvent(next, ctx->switch_event, 0));
    }
    ctx->stream = next;
    return 0;  // (the guard now records the event on the new stream: harmless, and it keeps "recorded" true)
}

extern "C" int mvlm_synchronize(mvlm_ctx* ctx) {
    MVLM_ENTER(ctx);
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// A device buffer for `room` bytes on the upload path (caller holds ctx->mu): the OLDEST pooled buffer that fits - its
// previous owner's work finished long ago, so the wait is a no-op and the copy really runs beside the current scan's
// kernels - or a new allocation.  `waited`: the previous owner's event (goes back to the event list with the new owner).
static bool take_upload_buffer(mvlm_ctx* ctx, size_t bytes, size_t room, void** p, size_t* cap, hipEvent_t* waited) {
    int pick = -1;
    for (int k = 0; k < int(ctx->mesh_pool.size()); ++k) {
        const size_t c = ctx->mesh_pool[size_t(k)].cap;
        if (c >= room && c <= 4 * bytes + (1u << 20)) {
            pick = k;
            break;
        }
    }
    if (pick >= 0) {
        const auto e = ctx->mesh_pool[size_t(pick)];
        ctx->mesh_pool.erase(ctx->mesh_pool.begin() + pick);
        ctx->mesh_pool_bytes -= e.cap;
        *p = e.p;
        *cap = e.cap;
        *waited = e.freed;
        return !(e.freed && hipStreamWaitEvent(ctx->upload_stream, e.freed, 0) != hipSuccess);
    }
    *cap = (room + 65535) / 65536 * 65536;
    if (hipMalloc(p, *cap) != hipSuccess) {
        *p = nullptr;
        *cap = 0;
        return false;
    }
    return true;
}

// a texture decoded ahead of its mesh (mvlm_texture_from_jpeg): the buffer a mesh upload takes over
struct mvlm_texture {
    uint8_t* dev = nullptr;
    size_t cap = 0;
    int h = 0, w = 0;
    hipEvent_t waited = nullptr;
};

// jpeg != null: the texture is decoded on the device from the staged JPEG (jpeg.hip) instead of copied from tex_host;
// pre != null: the texture is already on the device (mvlm_texture_from_jpeg) and its buffer becomes the mesh's;
// returns 2 (and no mesh) when the stream turns out not to decode - the caller then decodes on the host
static int mesh_upload_impl(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts, const int32_t* tris_host,
                            int n_tris, const uint8_t* tex_host, int tex_h, int tex_w, MvlmJpegPlan* jpeg,
                            const uint8_t* jpeg_bytes, size_t jpeg_n, mvlm_mesh** out, mvlm_texture* pre = nullptr) {
    // Called from reader threads while another thread launches kernels on this context.  The host work - index
    // validation (O(3T)), waiting for a staging slot, a possible hipHostMalloc, the 10-25 MB memcpy into pinned memory -
    // runs under the upload mutex only; the context mutex every launch entry point takes is held just for the pool
    // pick, the asynchronous copies and the event records.
    MVLM_REQUIRE(ctx, out, "mesh_upload: null output");
    *out = nullptr;
    MVLM_REQUIRE(ctx, verts_host && tris_host && n_verts > 0 && n_tris > 0, "mesh_upload: mesh does not contain any points");
    MVLM_REQUIRE(ctx, !(tex_host || jpeg || pre) || (tex_h > 0 && tex_w > 0), "mesh_upload: bad texture size");
    for (long i = 0; i < 3l * n_tris; ++i)
        MVLM_REQUIRE(ctx, tris_host[i] >= 0 && tris_host[i] < n_verts, "mesh_upload: triangle index out of range");
    const bool with_tex = (tex_host || jpeg || pre) && uvs_host;
    if (!with_tex) jpeg = nullptr;
    if (!with_tex) pre = nullptr;  // (no texture coordinates: the texture is not used - utils3d.py:26; the caller keeps it)
    const void* src[4] = {verts_host, uvs_host, tris_host, with_tex && !jpeg && !pre ? tex_host : nullptr};
    const size_t bytes[4] = {size_t(n_verts) * 12, uvs_host ? size_t(n_verts) * 8 : 0, size_t(n_tris) * 12,
                             with_tex ? size_t(tex_h) * tex_w * 3 : 0};
    // the rasteriser fetches a texel with one 4-byte load at byte 3 * index: 4 spare bytes behind the texture
    const size_t room[4] = {bytes[0], bytes[1], bytes[2], bytes[3] ? bytes[3] + 4 : 0};
    size_t off[5], total = 0;
    for (int i = 0; i < 4; ++i) {
        off[i] = total;
        total += ((src[i] ? bytes[i] : 0) + 255) / 256 * 256;
    }
    off[4] = total;  // the JPEG's header, tables and unstuffed stream
    if (jpeg) total += mvlm_jpeg_stage_bytes(*jpeg, jpeg_n);

    std::lock_guard<std::mutex> upload_lock(ctx->upload_mu);
    MVLM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    // the pinned staging slot (two slots alternate; a slot is reused once the copies out of it are done)
    if (!ctx->upload_stream) MVLM_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    const int slot = ctx->upload_stage_next;
    ctx->upload_stage_next ^= 1;
    if (ctx->upload_stage_done[slot]) MVLM_CHECK_HIP(ctx, hipEventSynchronize(ctx->upload_stage_done[slot]));
    if (ctx->upload_stage_cap[slot] < total) {
        if (ctx->upload_stage[slot]) (void)hipHostFree(ctx->upload_stage[slot]);
        ctx->upload_stage[slot] = nullptr;
        ctx->upload_stage_cap[slot] = 0;
        const size_t cap = (total + (size_t(4) << 20)) / (size_t(4) << 20) * (size_t(4) << 20);
        MVLM_CHECK_HIP(ctx, hipHostMalloc(&ctx->upload_stage[slot], cap, hipHostMallocDefault));
        ctx->upload_stage_cap[slot] = cap;
    }
    auto* stage = static_cast<unsigned char*>(ctx->upload_stage[slot]);
    for (int i = 0; i < 4; ++i)
        if (src[i] && bytes[i]) std::memcpy(stage + off[i], src[i], bytes[i]);
    std::string jpeg_why;
    if (jpeg && mvlm_jpeg_fill_stage(*jpeg, jpeg_bytes, jpeg_n, stage + off[4], jpeg_why) != 0) {
        ctx->fail("jpeg: " + jpeg_why);
        return 2;
    }

    auto* m = new mvlm_mesh();
    static std::atomic<unsigned long long> next_uid{1};
    m->uid = next_uid.fetch_add(1);
    m->n_verts = n_verts;
    m->n_tris = n_tris;
    void** dst[4] = {(void**)&m->verts, (void**)&m->uvs, (void**)&m->tris, (void**)&m->tex};
    bool ok = true;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);  // pool, event list: shared with the launch entry points
        for (int i = 0; i < 4 && ok; ++i) {
            if (!bytes[i]) continue;
            if (i == 3 && pre) {  // decoded ahead: the texture's buffer becomes the mesh's
                *dst[i] = pre->dev;
                m->cap[i] = pre->cap;
                m->waited[i] = pre->waited;
                continue;
            }
            if (!take_upload_buffer(ctx, bytes[i], room[i], dst[i], &m->cap[i], &m->waited[i])) ok = false;
            if (ok && src[i] && hipMemcpyAsync(*dst[i], stage + off[i], bytes[i], hipMemcpyHostToDevice, ctx->upload_stream) != hipSuccess) ok = false;
        }
        if (with_tex) {
            m->tex_h = tex_h;
            m->tex_w = tex_w;
        }
    }
    int jpeg_rc = 0;
    if (ok && jpeg) {
        // decode kernels on the upload stream; the host waits for that stream twice (jpeg.hip), with only the upload mutex
        // held - the launch thread keeps going
        jpeg_rc = mvlm_jpeg_run(ctx, *jpeg, stage + off[4], m->tex, ctx->upload_stream, jpeg_why, nullptr);
        if (jpeg_rc == 2) ctx->fail("jpeg: " + jpeg_why);
        if (jpeg_rc != 0) ok = false;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (ok) {
            if (!ctx->upload_stage_done[slot]) ctx->upload_stage_done[slot] = ctx->take_event();
            m->ready = ctx->take_event();
            ok = ctx->upload_stage_done[slot] && m->ready &&
                 hipEventRecord(ctx->upload_stage_done[slot], ctx->upload_stream) == hipSuccess &&
                 hipEventRecord(m->ready, ctx->upload_stream) == hipSuccess;
        }
    }
    if (!ok) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        if (pre) {  // still the caller's
            m->tex = nullptr;
            m->waited[3] = nullptr;
        }
        mvlm_mesh_free(nullptr, m);
        if (jpeg_rc != 0) return jpeg_rc;  // (the message is set)
        return ctx->fail("mesh_upload: device allocation / copy failed");
    }
    if (pre) delete pre;  // consumed: buffer and event are the mesh's now
    *out = m;
    return 0;
}

extern "C" int mvlm_mesh_upload(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                                const int32_t* tris_host, int n_tris, const uint8_t* tex_host, int tex_h, int tex_w,
                                mvlm_mesh** out) {
    return mesh_upload_impl(ctx, verts_host, uvs_host, n_verts, tris_host, n_tris, tex_host, tex_h, tex_w, nullptr, nullptr, 0, out);
}

// ---- JPEG textures decoded on the device (jpeg.hip) ------------------------------------------------------------------
extern "C" int mvlm_jpeg_info(const uint8_t* jpeg, size_t n_bytes, int* width, int* height, int* components, char* why, int why_len) {
    MvlmJpegPlan* plan = mvlm_jpeg_plan_new();
    std::string w;
    const int rc = mvlm_jpeg_plan_impl(jpeg, n_bytes, *plan, w);
    if (rc == 0) mvlm_jpeg_plan_dims(*plan, width, height, components);
    if (why && why_len > 0) {
        std::snprintf(why, size_t(why_len), "%s", w.c_str());
    }
    mvlm_jpeg_plan_delete(plan);
    return rc;
}

extern "C" int mvlm_mesh_upload_jpeg(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                                     const int32_t* tris_host, int n_tris, const uint8_t* jpeg, size_t jpeg_bytes, mvlm_mesh** out) {
    MVLM_REQUIRE(ctx, out, "mesh_upload_jpeg: null output");
    *out = nullptr;
    MVLM_REQUIRE(ctx, jpeg && jpeg_bytes > 0, "mesh_upload_jpeg: no JPEG");
    std::unique_ptr<MvlmJpegPlan, void (*)(MvlmJpegPlan*)> plan(mvlm_jpeg_plan_new(), mvlm_jpeg_plan_delete);
    std::string w;
    if (mvlm_jpeg_plan_impl(jpeg, jpeg_bytes, *plan, w) != 0) {
        ctx->fail("jpeg: " + w);
        return 2;
    }
    int tw = 0, th = 0, nc = 0;
    mvlm_jpeg_plan_dims(*plan, &tw, &th, &nc);
    return mesh_upload_impl(ctx, verts_host, uvs_host, n_verts, tris_host, n_tris, nullptr, th, tw, plan.get(), jpeg, jpeg_bytes, out);
}

// the decoder alone (tests, tools): rgb_dev [H, W, 3] is complete on return; rounds_out: synchronisation rounds it took
extern "C" int mvlm_jpeg_decode(mvlm_ctx* ctx, const uint8_t* jpeg, size_t jpeg_bytes, uint8_t* rgb_dev, int* rounds_out) {
    MVLM_REQUIRE(ctx, jpeg && jpeg_bytes > 0 && rgb_dev, "jpeg_decode: null pointer");
    std::unique_ptr<MvlmJpegPlan, void (*)(MvlmJpegPlan*)> plan(mvlm_jpeg_plan_new(), mvlm_jpeg_plan_delete);
    std::string w;
    if (mvlm_jpeg_plan_impl(jpeg, jpeg_bytes, *plan, w) != 0) {
        ctx->fail("jpeg: " + w);
        return 2;
    }
    std::lock_guard<std::mutex> upload_lock(ctx->upload_mu);
    MVLM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->upload_stream) MVLM_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    const size_t total = mvlm_jpeg_stage_bytes(*plan, jpeg_bytes);
    const int slot = ctx->upload_stage_next;
    ctx->upload_stage_next ^= 1;
    if (ctx->upload_stage_done[slot]) MVLM_CHECK_HIP(ctx, hipEventSynchronize(ctx->upload_stage_done[slot]));
    if (ctx->upload_stage_cap[slot] < total) {
        if (ctx->upload_stage[slot]) (void)hipHostFree(ctx->upload_stage[slot]);
        ctx->upload_stage[slot] = nullptr;
        ctx->upload_stage_cap[slot] = 0;
        const size_t cap = (total + (size_t(4) << 20)) / (size_t(4) << 20) * (size_t(4) << 20);
        MVLM_CHECK_HIP(ctx, hipHostMalloc(&ctx->upload_stage[slot], cap, hipHostMallocDefault));
        ctx->upload_stage_cap[slot] = cap;
    }
    auto* stage = static_cast<unsigned char*>(ctx->upload_stage[slot]);
    if (mvlm_jpeg_fill_stage(*plan, jpeg, jpeg_bytes, stage, w) != 0) {
        ctx->fail("jpeg: " + w);
        return 2;
    }
    // whatever the launch stream still does with rgb_dev comes first
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        hipEvent_t e = ctx->take_event();
        MVLM_REQUIRE(ctx, e, "jpeg_decode: no event");
        const bool ok = hipEventRecord(e, ctx->stream) == hipSuccess && hipStreamWaitEvent(ctx->upload_stream, e, 0) == hipSuccess;
        ctx->event_free.push_back(e);
        MVLM_REQUIRE(ctx, ok, "jpeg_decode: stream order");
    }
    const int rc = mvlm_jpeg_run(ctx, *plan, stage, rgb_dev, ctx->upload_stream, w, rounds_out);
    if (rc == 2) ctx->fail("jpeg: " + w);
    return rc;  // (mvlm_jpeg_run has waited for the stream: the staging slot is free, the image complete)
}

// The texture of a scan decoded AHEAD of its mesh, so that a caller can decode (GPU + this thread waiting for it) while
// another thread still parses the geometry: the JPEG goes through the decoder into a buffer of the mesh pool; the handle is
// consumed by mvlm_mesh_upload_texture or given back with mvlm_texture_free.  0 / 1 / 2 like mvlm_mesh_upload_jpeg.
extern "C" int mvlm_texture_from_jpeg(mvlm_ctx* ctx, const uint8_t* jpeg, size_t jpeg_bytes, mvlm_texture** out) {
    MVLM_REQUIRE(ctx, out, "texture_from_jpeg: null output");
    *out = nullptr;
    MVLM_REQUIRE(ctx, jpeg && jpeg_bytes > 0, "texture_from_jpeg: no JPEG");
    std::unique_ptr<MvlmJpegPlan, void (*)(MvlmJpegPlan*)> plan(mvlm_jpeg_plan_new(), mvlm_jpeg_plan_delete);
    std::string w;
    if (mvlm_jpeg_plan_impl(jpeg, jpeg_bytes, *plan, w) != 0) {
        ctx->fail("jpeg: " + w);
        return 2;
    }
    int tw = 0, th = 0, nc = 0;
    mvlm_jpeg_plan_dims(*plan, &tw, &th, &nc);
    std::lock_guard<std::mutex> upload_lock(ctx->upload_mu);
    MVLM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->upload_stream) MVLM_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    const size_t total = mvlm_jpeg_stage_bytes(*plan, jpeg_bytes);
    const int slot = ctx->upload_stage_next;
    ctx->upload_stage_next ^= 1;
    if (ctx->upload_stage_done[slot]) MVLM_CHECK_HIP(ctx, hipEventSynchronize(ctx->upload_stage_done[slot]));
    if (ctx->upload_stage_cap[slot] < total) {
        if (ctx->upload_stage[slot]) (void)hipHostFree(ctx->upload_stage[slot]);
        ctx->upload_stage[slot] = nullptr;
        ctx->upload_stage_cap[slot] = 0;
        const size_t cap = (total + (size_t(4) << 20)) / (size_t(4) << 20) * (size_t(4) << 20);
        MVLM_CHECK_HIP(ctx, hipHostMalloc(&ctx->upload_stage[slot], cap, hipHostMallocDefault));
        ctx->upload_stage_cap[slot] = cap;
    }
    auto* stage = static_cast<unsigned char*>(ctx->upload_stage[slot]);
    if (mvlm_jpeg_fill_stage(*plan, jpeg, jpeg_bytes, stage, w) != 0) {
        ctx->fail("jpeg: " + w);
        return 2;
    }
    std::unique_ptr<mvlm_texture> t(new mvlm_texture());
    t->h = th;
    t->w = tw;
    const size_t bytes = size_t(th) * tw * 3;
    bool ok;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        void* p = nullptr;
        ok = take_upload_buffer(ctx, bytes, bytes + 4, &p, &t->cap, &t->waited);  // (4 spare bytes: the rasteriser's texel load)
        t->dev = static_cast<uint8_t*>(p);
    }
    int rc = ok ? mvlm_jpeg_run(ctx, *plan, stage, t->dev, ctx->upload_stream, w, nullptr) : 1;
    if (rc != 0) {
        if (rc == 2) ctx->fail("jpeg: " + w);
        if (!ok) ctx->fail("texture_from_jpeg: device allocation failed");
        mvlm_texture_free(ctx, t.release());
        return rc;
    }
    *out = t.release();  // (mvlm_jpeg_run has waited for the stream: the pixels are there)
    return 0;
}

extern "C" void mvlm_texture_free(mvlm_ctx* ctx, mvlm_texture* t) {
    if (!t) return;
    if (ctx && t->dev) {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (t->waited) ctx->event_free.push_back(t->waited);
        t->waited = nullptr;
        if (ctx->mesh_pool.size() < 32 && ctx->mesh_pool_bytes + t->cap <= (size_t(1) << 30)) {
            // only the upload stream has touched the buffer, and the decode waited for it: no event needed
            ctx->mesh_pool.push_back({t->dev, t->cap, nullptr});
            ctx->mesh_pool_bytes += t->cap;
            t->dev = nullptr;
        }
    }
    if (t->dev || t->waited) {
        if (t->waited) (void)hipEventDestroy(t->waited);
        if (t->dev) (void)hipFree(t->dev);
    }
    delete t;
}

extern "C" int mvlm_texture_size(const mvlm_texture* t, int* height, int* width) {
    if (!t) return 1;
    if (height) *height = t->h;
    if (width) *width = t->w;
    return 0;
}

// mvlm_mesh_upload with a texture that is on the device already; on success the handle is consumed (do not free it), on
// failure it is still the caller's.  Without texture coordinates the texture is not used and stays the caller's too.
extern "C" int mvlm_mesh_upload_texture(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                                        const int32_t* tris_host, int n_tris, mvlm_texture* tex, int* consumed, mvlm_mesh** out) {
    if (consumed) *consumed = 0;
    MVLM_REQUIRE(ctx, tex && tex->dev, "mesh_upload_texture: no texture");
    const bool will_consume = uvs_host != nullptr;
    const int rc = mesh_upload_impl(ctx, verts_host, uvs_host, n_verts, tris_host, n_tris, nullptr, tex->h, tex->w, nullptr, nullptr, 0, out, tex);
    if (rc == 0 && will_consume && consumed) *consumed = 1;
    return rc;
}

// Buffers go back to the context's pool together with an event on the launch stream: whatever was enqueued for this
// mesh before the free may still read them.  ctx == NULL (or a pool that is full): plain hipFree (which waits for the device).
extern "C" void mvlm_mesh_free(mvlm_ctx* ctx, mvlm_mesh* m) {
    if (!m) return;
    void* bufs[4] = {m->verts, m->uvs, m->tris, m->tex};
    constexpr size_t POOL_MAX_BYTES = size_t(1) << 30;
    constexpr size_t POOL_MAX_ENTRIES = 32;
    hipEvent_t spare[5] = {m->ready, m->waited[0], m->waited[1], m->waited[2], m->waited[3]};
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->mu);
        (void)hipSetDevice(ctx->device);
        for (int i = 0; i < 4; ++i)
            if (bufs[i] && m->cap[i] && ctx->mesh_pool.size() < POOL_MAX_ENTRIES &&
                ctx->mesh_pool_bytes + m->cap[i] <= POOL_MAX_BYTES) {
                hipEvent_t ev = ctx->take_event();
                if (!ev || hipEventRecord(ev, ctx->stream) != hipSuccess) {  // no event: keep the plain (synchronising) free
                    if (ev) ctx->event_free.push_back(ev);
                    continue;
                }
                ctx->mesh_pool.push_back({bufs[i], m->cap[i], ev});
                ctx->mesh_pool_bytes += m->cap[i];
                bufs[i] = nullptr;
            }
        for (hipEvent_t e : spare)
            if (e) ctx->event_free.push_back(e);
    } else {
        for (hipEvent_t e : spare)
            if (e) (void)hipEventDestroy(e);
    }
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete m;
}

// test / tuning hook: the next mvlm_conv2d calls on this ctx use exactly this kernel variant (-1: the dispatcher's choice)
extern "C" int mvlm_conv_force_variant(mvlm_ctx* ctx, int variant) {
    MVLM_ENTER(ctx);
    ctx->conv_force_variant = variant < 0 ? -1 : variant;
    return 0;
}

// ---- single convolution (test hook): packs the weights like mvlm_amd/weights.py does -------
extern "C" int mvlm_conv2d(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host,
                           int cout, int ksize, const float* bias_host, const float* pre_scale_host,
                           const float* pre_shift_host, const float* post_scale_host, const float* post_shift_host,
                           const float* r_dev, int upsample_in, float* y_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, x_dev && w_host && y_dev, "conv2d: null pointer");
    MVLM_REQUIRE(ctx, ksize == 1 || ksize == 3, "conv2d: kernel size must be 1 or 3");
    MVLM_REQUIRE(ctx, (pre_scale_host == nullptr) == (pre_shift_host == nullptr), "conv2d: pre scale/shift come in pairs");
    MVLM_REQUIRE(ctx, (post_scale_host == nullptr) == (post_shift_host == nullptr), "conv2d: post scale/shift come in pairs");
    // same padding rules as mvlm_amd/weights.py: a plain conv + bias layer on the 32-pixel-row tiles may
    // end in one 16-row strip
    const bool tail16 = ksize == 3 && (cout + 15) / 16 * 16 == 80 && !pre_scale_host && !post_scale_host && !r_dev &&
                        w >= 32 && h % 8 == 0;
    const int cin_pad = (ksize == 1 ? (cin + 7) / 8 * 8 : (cin + 3) / 4 * 4), taps = ksize * ksize;
    // ... or, with exactly 84 output channels, in a 16-row and a 4-row strip (conv6 / conv10 of the 84-landmark network)
    const bool tail4 = ksize == 3 && cout == 84 && bias_host && !pre_scale_host && !post_scale_host && !r_dev && !upsample_in &&
                       w >= 32 && h % 8 == 0;
    const int cout_pad = tail16 ? (cout + 15) / 16 * 16 : tail4 ? 84 : (cout + 31) / 32 * 32;
    std::vector<float> blob;
    auto push = [&](size_t n) {
        const size_t off = blob.size();
        blob.resize(off + (n + 3) / 4 * 4, 0.f);
        return off;
    };
    const size_t w_off = push(size_t(taps) * cin_pad * cout_pad);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t)
                blob[w_off + (size_t(t) * cin_pad + ci) * cout_pad + co] = w_host[(size_t(co) * cin + ci) * taps + t];
    auto vec = [&](const float* src, int n, int n_pad) -> long {
        if (!src) return -1;
        const size_t off = push(n_pad);
        for (int i = 0; i < n; ++i) blob[off + i] = src[i];
        return long(off);
    };
    const long b_off = vec(bias_host, cout, cout_pad);
    const long ps_off = vec(pre_scale_host, cin, cin_pad), pt_off = vec(pre_shift_host, cin, cin_pad);
    const long qs_off = vec(post_scale_host, cout, cout_pad), qt_off = vec(post_shift_host, cout, cout_pad);
    auto* d = static_cast<float*>(ctx->get_scratch("conv2d.blob", blob.size() * 4));
    MVLM_REQUIRE(ctx, d, "conv2d: scratch allocation failed");
    MVLM_CHECK_HIP(ctx, hipMemcpy(d, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = x_dev;
    a.in_ctot = cin;
    a.cin = cin;
    a.cin_pad = cin_pad;
    a.up_in = upsample_in;
    a.B = batch;
    a.H = h;
    a.W = w;
    a.w = d + w_off;
    a.cout = cout;
    a.cout_pad = cout_pad;
    a.ksize = ksize;
    a.bias = b_off < 0 ? nullptr : d + b_off;
    a.pre_scale = ps_off < 0 ? nullptr : d + ps_off;
    a.pre_shift = pt_off < 0 ? nullptr : d + pt_off;
    a.post_scale = qs_off < 0 ? nullptr : d + qs_off;
    a.post_shift = qt_off < 0 ? nullptr : d + qt_off;
    a.res1 = r_dev;
    a.res1_ctot = cout;
    a.out = y_dev;
    a.out_ctot = cout;
    if (mvlm_launch_conv(ctx, a, nullptr)) return 1;
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- kernel-variant timing (tools/tune_conv.py): one layer shape, dummy data, a forced variant ----------
// Times `iters` launches of one convolution with HIP events on the ctx stream.  flags: 1 = pre-BN+ReLU on the
// input, 2 = residual add + raw copy (a residual block's conv1 / conv2), 4 = bias, 8 = post-BN+ReLU.
// variant -1: the dispatcher's own choice, -2: its rules without the tuned table (returned in *variant_used).
// Shapes a variant cannot serve fail.
extern "C" int mvlm_conv_bench(mvlm_ctx* ctx, int batch, int cin, int cout, int ksize, int size, int flags, int variant,
                               int iters, float* ms_per_launch, int* variant_used) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, batch > 0 && cin > 0 && cout > 0 && size > 0 && iters > 0 && ms_per_launch, "conv_bench: bad arguments");
    MVLM_REQUIRE(ctx, ksize == 1 || ksize == 3, "conv_bench: kernel size must be 1 or 3");
    const int cin_pad = (ksize == 1 ? (cin + 7) / 8 * 8 : (cin + 3) / 4 * 4), taps = ksize * ksize;
    const bool plain = (flags & 4) && !(flags & (1 | 2 | 8));
    const int cout_pad = (plain && (cout + 15) / 16 * 16 == 80) ? 80 : (plain && ksize == 3 && cout == 84) ? 84 : (cout + 31) / 32 * 32;
    const size_t px = size_t(batch) * size * size;
    const size_t n_w = size_t(taps) * cin_pad * cout_pad, n_vec = size_t(cin_pad) * 2 + size_t(cout_pad) * 3;
    const size_t n_x = px * cin, n_y = px * cout;
    const size_t total = n_w + n_vec + n_x + 3 * n_y + 64;
    auto* base = static_cast<float*>(ctx->get_scratch("conv_bench", total * sizeof(float)));
    MVLM_REQUIRE(ctx, base, "conv_bench: scratch allocation failed");
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(base, 0, total * sizeof(float), ctx->stream));
    float* w = base;
    float* vec = w + n_w;
    float* x = vec + (n_vec + 3) / 4 * 4;
    float* y = x + (n_x + 3) / 4 * 4;
    float* raw = y + (n_y + 3) / 4 * 4;
    float* res = raw + (n_y + 3) / 4 * 4;
    ConvArgs a;
    a.in = x;
    a.in_ctot = cin;
    a.cin = cin;
    a.cin_pad = cin_pad;
    a.B = batch;
    a.H = a.W = size;
    a.w = w;
    a.cout = cout;
    a.cout_pad = cout_pad;
    a.ksize = ksize;
    if (flags & 1) {
        a.pre_scale = vec;
        a.pre_shift = vec + cin_pad;
    }
    if (flags & 4) a.bias = vec + 2 * cin_pad;
    if (flags & 8) {
        a.post_scale = vec + 2 * cin_pad + cout_pad;
        a.post_shift = vec + 2 * cin_pad + 2 * cout_pad;
    }
    if (flags & 2) {
        a.res1 = res;
        a.res1_ctot = cout;
        a.out_raw = raw;
        a.raw_ctot = cout;
    }
    a.out = y;
    a.out_ctot = cout;
    const int saved = ctx->conv_force_variant;
    ctx->conv_force_variant = variant >= 0 ? variant : (variant == -2 ? -2 : -1);
    int used = -1;
    unsigned short* wq = nullptr;
    // the opt-in split kernels on zero weights: 62 bf16x3, 61 f16x2
    const int splits = variant == MVLM_CONV_VARIANT_FAST ? 3 : (variant == MVLM_CONV_VARIANT_FAST16 ? 2 : 0);
    if (splits) {
        a.cin_pad = (cin + 15) / 16 * 16;
        a.cout_pad = mvlm_fast_cout_pad(cout);
        const size_t n16 = size_t(a.cin_pad / 16) * 9 * 2 * splits * a.cout_pad * 8;
        wq = static_cast<unsigned short*>(ctx->get_scratch("conv_bench.wq", n16 * 2));
        if (!wq || !mvlm_conv_fast_ok(a, splits)) {
            ctx->conv_force_variant = saved;
            return ctx->fail("conv_bench: shape not eligible for the fast kernel");
        }
        (void)hipMemsetAsync(wq, 0, n16 * 2, ctx->stream);
    }
    auto launch = [&](int* v) { return wq ? mvlm_launch_conv_fast(ctx, a, wq, splits, 1.f) : mvlm_launch_conv(ctx, a, v); };
    int rc = launch(&used);  // warm-up (also sets the launch attributes)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = ctx->fail("conv_bench: hipEventCreate failed");
    if (!rc) {
        hipEventRecord(e0, ctx->stream);
        for (int i = 0; i < iters && !rc; ++i) rc = launch(nullptr);
        hipEventRecord(e1, ctx->stream);
        if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = ctx->fail("conv_bench: kernel failed");
        float ms = 0.f;
        if (!rc) hipEventElapsedTime(&ms, e0, e1);
        *ms_per_launch = ms / iters;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    ctx->conv_force_variant = saved;
    if (variant_used) *variant_used = used;
    return rc;
}

// ---- two independent 3x3 convolutions in one launch (test hook for conv_pair_kernel) -----------------------------------
// Problem i: x_i f32[B,cin,size_i,size_i] * w_i f32[cout,cin,3,3] (+ shared pre-BN+ReLU, + residual r_i, raw copy raw_i)
// -> y_i; `variant` as in mvlm_conv_pair_bench.  The results must equal mvlm_conv2d's with the same kernel variant forced.
extern "C" int mvlm_conv2d_pair(mvlm_ctx* ctx, int batch, int cin, int cout, const float* x0_dev, int size0, const float* w0_host,
                                const float* r0_dev, float* raw0_dev, float* y0_dev, const float* x1_dev, int size1,
                                const float* w1_host, const float* r1_dev, float* raw1_dev, float* y1_dev,
                                const float* pre_scale_host, const float* pre_shift_host, int variant) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, x0_dev && x1_dev && w0_host && w1_host && y0_dev && y1_dev && variant >= 0, "conv2d_pair: bad arguments");
    MVLM_REQUIRE(ctx, (pre_scale_host == nullptr) == (pre_shift_host == nullptr), "conv2d_pair: pre scale/shift come in pairs");
    const int cin_pad = (cin + 3) / 4 * 4, cout_pad = (cout + 31) / 32 * 32;
    const size_t n_w = (size_t(9) * cin_pad * cout_pad + 3) / 4 * 4;
    std::vector<float> blob(2 * n_w + 2 * size_t(cin_pad), 0.f);
    const float* ws[2] = {w0_host, w1_host};
    for (int i = 0; i < 2; ++i)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) blob[i * n_w + (size_t(t) * cin_pad + ci) * cout_pad + co] = ws[i][(size_t(co) * cin + ci) * 9 + t];
    if (pre_scale_host)
        for (int i = 0; i < cin; ++i) {
            blob[2 * n_w + i] = pre_scale_host[i];
            blob[2 * n_w + cin_pad + i] = pre_shift_host[i];
        }
    auto* d = static_cast<float*>(ctx->get_scratch("conv2d.blob", blob.size() * 4));
    MVLM_REQUIRE(ctx, d, "conv2d_pair: scratch allocation failed");
    MVLM_CHECK_HIP(ctx, hipMemcpy(d, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a[2];
    const float* xs[2] = {x0_dev, x1_dev};
    const float* rs[2] = {r0_dev, r1_dev};
    float* raws[2] = {raw0_dev, raw1_dev};
    float* ys[2] = {y0_dev, y1_dev};
    const int sizes[2] = {size0, size1};
    for (int i = 0; i < 2; ++i) {
        ConvArgs& c = a[i];
        c.in = xs[i];
        c.in_ctot = cin;
        c.cin = cin;
        c.cin_pad = cin_pad;
        c.B = batch;
        c.H = c.W = sizes[i];
        c.w = d + i * n_w;
        c.cout = cout;
        c.cout_pad = cout_pad;
        c.ksize = 3;
        if (pre_scale_host) {
            c.pre_scale = d + 2 * n_w;
            c.pre_shift = d + 2 * n_w + cin_pad;
        }
        c.res1 = rs[i];
        c.res1_ctot = cout;
        c.out_raw = raws[i];
        c.raw_ctot = cout;
        c.out = ys[i];
        c.out_ctot = cout;
    }
    if (mvlm_launch_conv_pair(ctx, a[0], a[1], (variant & 0xfff) | MVLM_CONV_PAIR_FLAG)) return 1;
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Tuning hook for tools/tune_conv_pairs.py: conv j of a hourglass level's skip block at `size` and of the next level's first
// block at size / 2 (same channels, pre-BN + residual + raw copy as in the network) on zero data.  variant < 0: the two
// tuned single launches back to back; variant >= 0: ONE two-problem launch of kernel variant (variant & 255) with
// 1 << ((variant >> 8) & 3) / 1 << ((variant >> 10) & 3) K parts for the size / size / 2 problem.
extern "C" int mvlm_conv_pair_bench(mvlm_ctx* ctx, int batch, int cin, int cout, int size, int flags, int variant, int iters,
                                    float* ms_per_launch) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, batch > 0 && cin > 0 && cout > 0 && size >= 8 && size % 2 == 0 && iters > 0 && ms_per_launch, "conv_pair_bench: bad arguments");
    const int cin_pad = (cin + 3) / 4 * 4, cout_pad = (cout + 31) / 32 * 32;
    const size_t n_w = size_t(9) * cin_pad * cout_pad, n_vec = (size_t(cin_pad) * 2 + size_t(cout_pad) * 3 + 3) / 4 * 4;
    size_t px[2] = {size_t(batch) * size * size, size_t(batch) * (size / 2) * (size / 2)};
    size_t total = n_w + n_vec + 64;
    for (int i = 0; i < 2; ++i) total += (px[i] * cin + 3) / 4 * 4 + 3 * ((px[i] * cout + 3) / 4 * 4);
    auto* base = static_cast<float*>(ctx->get_scratch("conv_bench", total * sizeof(float)));
    MVLM_REQUIRE(ctx, base, "conv_pair_bench: scratch allocation failed");
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(base, 0, total * sizeof(float), ctx->stream));
    float* w = base;
    float* vec = w + n_w;
    float* cur = vec + n_vec;
    ConvArgs a[2];
    for (int i = 0; i < 2; ++i) {
        ConvArgs& c = a[i];
        const size_t n_x = (px[i] * cin + 3) / 4 * 4, n_y = (px[i] * cout + 3) / 4 * 4;
        c.in = cur;
        float* y = cur + n_x;
        float* raw = y + n_y;
        float* res = raw + n_y;
        cur = res + n_y;
        c.in_ctot = cin;
        c.cin = cin;
        c.cin_pad = cin_pad;
        c.B = batch;
        c.H = c.W = i ? size / 2 : size;
        c.w = w;
        c.cout = cout;
        c.cout_pad = cout_pad;
        c.ksize = 3;
        if (flags & 1) {
            c.pre_scale = vec;
            c.pre_shift = vec + cin_pad;
        }
        if (flags & 2) {
            c.res1 = res;
            c.res1_ctot = cout;
            c.out_raw = raw;
            c.raw_ctot = cout;
        }
        c.out = y;
        c.out_ctot = cout;
    }
    const int saved = ctx->conv_force_variant;
    ctx->conv_force_variant = -1;
    auto launch = [&]() -> int {
        if (variant < 0) return mvlm_launch_conv(ctx, a[0], nullptr) || mvlm_launch_conv(ctx, a[1], nullptr);
        return mvlm_launch_conv_pair(ctx, a[0], a[1], (variant & 0xfff) | MVLM_CONV_PAIR_FLAG);
    };
    int rc = launch();  // warm-up (also sets the launch attributes)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = ctx->fail("conv_pair_bench: hipEventCreate failed");
    if (!rc) {
        hipEventRecord(e0, ctx->stream);
        for (int i = 0; i < iters && !rc; ++i) rc = launch();
        hipEventRecord(e1, ctx->stream);
        if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = ctx->fail("conv_pair_bench: kernel failed");
        float ms = 0.f;
        if (!rc) hipEventElapsedTime(&ms, e0, e1);
        *ms_per_launch = ms / iters;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    ctx->conv_force_variant = saved;
    return rc;
}
