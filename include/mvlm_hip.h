/*
 * mvlm_hip.h - C ABI of libmvlm_hip.so: the MI355X (gfx950) implementation of the
 * cvjena/mvlm `Pipeline.predict_one_file` hot path.
 *
 * The reference is pure Python and has no FFI of its own; the interface each entry
 * point replaces is the Python method cited next to it (paths relative to the
 * reference's src/mvlm/).  INTEGRATION.md shows the ctypes binding a maintainer
 * adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from mvlm_last_error(ctx) - the CALLING THREAD's last failure on that
 *     ctx, valid until that thread's next failing call.  Nothing throws across the boundary.
 *   - "_dev" pointers are device (HBM) addresses owned by the caller (e.g. a torch
 *     tensor's data_ptr()); "_host" pointers are ordinary host memory.  The library
 *     never frees caller memory.
 *   - all work is enqueued on the ctx's HIP stream (mvlm_set_stream; default: the
 *     null stream).  Functions that return host results synchronise that stream.
 *   - a ctx serialises its callers with an internal mutex (the reference server
 *     calls predict_one_file from a thread pool without locks, 3DMD_server.py:26-31);
 *     one C call is atomic, a SEQUENCE of calls that shares buffers is the caller's to
 *     order (the Python Pipeline holds a lock of its own around a whole scan).
 *     The mvlm_mesh_upload* / mvlm_texture_* / mvlm_jpeg_decode calls may run on other threads
 *     beside the launching one: their host work and their device work (a copy / decode stream
 *     of the context's own) happen outside the ctx mutex, under an upload mutex of their own.
 *     mvlm_obj_read / mvlm_mesh_read / mvlm_jpeg_info need no ctx at all
 *     (mvlm_obj_read parses on up to 16 threads of its own - the host's cores divided by the number of
 *     mvlm_obj_read calls in flight; MVLM_OBJ_THREADS overrides).
 */
#ifndef MVLM_HIP_H
#define MVLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvlm_ctx mvlm_ctx;
typedef struct mvlm_mesh mvlm_mesh;
typedef struct mvlm_texture mvlm_texture;

#define MVLM_IMAGE_SIZE 256          /* render3d.py / general_pipeline.py:57 */
#define MVLM_CONV_DESC_INTS 12       /* ints per conv slot, see mvlm_amd/weights.py */

enum { MVLM_MODE_QUANTILE = 0, MVLM_MODE_ABSOLUTE = 1 };   /* estimator3d.py:166-171 */
enum { MVLM_MAXIMA_SIMPLE = 0, MVLM_MAXIMA_MOMENT = 1 };   /* paulsenpredictor.py:121,129 */

/* ---- context ------------------------------------------------------------------ */
int mvlm_ctx_create(int device, mvlm_ctx** out);
void mvlm_ctx_destroy(mvlm_ctx* ctx);
const char* mvlm_last_error(mvlm_ctx* ctx);
/* the stream the context's next calls enqueue on (default: the null stream).  A context reuses device scratch from call to
 * call; when the stream CHANGES, the new stream is made to wait for what the context enqueued on the previous one, so two
 * users of one context under different streams stay ordered. */
int mvlm_set_stream(mvlm_ctx* ctx, void* hip_stream);
int mvlm_synchronize(mvlm_ctx* ctx);
/* name of the GPU architecture the library was built for ("gfx950") */
const char* mvlm_build_arch(void);

/* ---- OBJ ingest on the host (replaces utils3d.py:16-24, vtkOBJReader) ------------ */
/* Parses a Wavefront OBJ into float32 points, one texture coordinate per point (a point is
 * duplicated when it is used with several `vt` indices), polygons as triangle fans; `.mtl`,
 * normals and groups are ignored.  Needs no ctx and no GPU.  Errors mirror the reference's
 * ValueErrors: missing file (utils3d.py:13-14), no points (:20-21). */
typedef struct mvlm_obj mvlm_obj;
enum { MVLM_OBJ_ERR_ARGS = 1, MVLM_OBJ_ERR_FILE = 2, MVLM_OBJ_ERR_EMPTY = 3, MVLM_OBJ_ERR_INDEX = 4,
       MVLM_OBJ_ERR_SYNTAX = 5 };
int mvlm_obj_read(const char* path, mvlm_obj** out, char* err, int err_len);
int mvlm_obj_info(const mvlm_obj* obj, int64_t* n_verts, int64_t* n_tris, int* has_uvs);
/* copies into caller arrays sized from mvlm_obj_info: verts f32[V,3], uvs f32[V,2] (may be NULL),
 * tris i32[T,3] */
int mvlm_obj_copy(const mvlm_obj* obj, float* verts, float* uvs, int32_t* tris);
void mvlm_obj_free(mvlm_obj* obj);
/* The reference's legacy multi-format reader (Utils3D.multi_read_surface, utils3d.py:389-423) by file extension:
 * .obj (as above), .ply (ASCII / binary), .stl (ASCII / binary, coincident points merged like vtkSTLReader),
 * .vtk (legacy POLYDATA, ASCII / BINARY), .wrl (VRML 2.0 IndexedFaceSet, the last one of the file).  Fills the
 * same handle; texture coordinates are kept where the format carries them per point / corner. */
int mvlm_mesh_read(const char* path, mvlm_obj** out, char* err, int err_len);

/* ---- mesh (replaces utils3d.py:10-85 obj_to_actor's upload half) ---------------- */
/* verts f32[V,3], uvs f32[V,2] or NULL, tris i32[T,3], tex u8[H,W,3] (row 0 = top of
 * the image file) or NULL (=> pure white mesh, utils3d.py:58-64).  Host pointers;
 * the data is copied into pinned staging before the call returns (the host arrays are
 * the caller's again) and travels to the device asynchronously on a copy stream of
 * the context's own; mvlm_render / mvlm_project_to_surface / mvlm_clip_rays_to_mesh
 * wait for it on their stream.  May be called from another thread than the one that
 * launches work (a reader thread uploading the next scan). */
int mvlm_mesh_upload(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                     const int32_t* tris_host, int n_tris, const uint8_t* tex_host, int tex_h, int tex_w,
                     mvlm_mesh** out);
void mvlm_mesh_free(mvlm_ctx* ctx, mvlm_mesh* mesh);

/* ---- JPEG texture decoded on the device (replaces vtkJPEGReader in obj_to_actor, utils3d.py:28-34 / :42-48, and in
 * multi_read_surface's texture lookup, utils3d.py:457-462; the decoder behind vtkJPEGReader is libjpeg-turbo with its
 * defaults - JDCT_ISLOW, fancy upsampling, RGB out - and the bytes produced here are libjpeg's) ---------------------
 * Taken: baseline / extended sequential Huffman JPEG (SOF0 / SOF1), 8 bit, one interleaved scan, grey or YCbCr with
 * 1x1 / 2x1 / 2x2 luma sampling over 1x1 chroma, restart intervals.  Return code 2 = "not taken" (progressive,
 * arithmetic, CMYK, 12 bit, other sampling, corrupt entropy-coded data; mvlm_last_error / why says which): the caller
 * decodes such a file with libjpeg on the host and uploads the pixels with mvlm_mesh_upload. */
/* header only, no GPU, no ctx: 0 and the size if the device decoder takes this stream, 2 and the reason in why if not */
int mvlm_jpeg_info(const uint8_t* jpeg_host, size_t n_bytes, int* width, int* height, int* components, char* why, int why_len);
/* mvlm_mesh_upload with the texture given as the JPEG file's bytes: headers are parsed and byte stuffing is removed on the
 * host (into pinned staging), entropy decoding, inverse DCT, chroma upsampling and colour conversion run on the context's
 * upload stream.  The calling thread waits for that stream (not for the launch stream) before it returns. */
int mvlm_mesh_upload_jpeg(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                          const int32_t* tris_host, int n_tris, const uint8_t* jpeg_host, size_t jpeg_bytes,
                          mvlm_mesh** out);
/* The texture decoded AHEAD of its mesh, so that one thread can decode (GPU work + this thread waiting for it) while another
 * still parses the geometry (obj_to_actor does both in sequence, utils3d.py:16-34): the pixels go into a device buffer of
 * the context's mesh pool.  The handle is consumed by a successful mvlm_mesh_upload_texture that uses it (*consumed = 1; a
 * mesh without texture coordinates does not use a texture, utils3d.py:26) and must be given back with mvlm_texture_free
 * otherwise.  Return codes as mvlm_mesh_upload_jpeg (2 = not taken: decode on the host). */
int mvlm_texture_from_jpeg(mvlm_ctx* ctx, const uint8_t* jpeg_host, size_t jpeg_bytes, mvlm_texture** out);
int mvlm_texture_size(const mvlm_texture* tex, int* height, int* width);
void mvlm_texture_free(mvlm_ctx* ctx, mvlm_texture* tex);
int mvlm_mesh_upload_texture(mvlm_ctx* ctx, const float* verts_host, const float* uvs_host, int n_verts,
                             const int32_t* tris_host, int n_tris, mvlm_texture* tex, int* consumed, mvlm_mesh** out);
/* the decoder alone: rgb_dev u8[H,W,3] (sizes from mvlm_jpeg_info) is complete when the call returns; rounds_out (may be
 * NULL): how many synchronisation rounds the parallel entropy decoder needed */
int mvlm_jpeg_decode(mvlm_ctx* ctx, const uint8_t* jpeg_host, size_t jpeg_bytes, uint8_t* rgb_dev, int* rounds_out);

/* ---- render (replaces render3d.py:114-177 + :191, all poses in one launch set) --- */
/* rot_host f64[N,9]: row-major M = Ry*Rx*Rz per view (render3d.py:140-144).
 * out_dev f32[N,256,256,4]: RGB + depth planes in [0,1], already flipped to
 * image orientation (render3d.py:177) and divided by 255 (:191). */
int mvlm_render(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* rot_host, int n_views, float* out_dev);
/* The device copy of the last render's rotations, f64[n_views,9] (the context's scratch: valid until the next mvlm_render):
 * mvlm_estimate_lines of the same views takes it as rot_dev, and the table crosses PCIe once per mesh. */
int mvlm_render_rotations_dev(mvlm_ctx* ctx, const double** rot_dev);
/* what the RGB planes hold: 0 = unlit white x nearest texel, exactly what the reference renders
 * (utils3d.py:58-64); 1 = build-defined "geometry" shading (flat two-sided head light, grey in all
 * three planes) for models trained on geometry renderings - the reference's renderer has no such
 * mode, parity unpinned. */
int mvlm_set_render_shading(mvlm_ctx* ctx, int shading);
/* The one rasterisation parameter OpenGL leaves to the implementation that is a NUMBER: vertices snap to 2^-bits pixel
 * (GL_SUBPIXEL_BITS; the reference's images depend on the OpenGL its VTK runs on, render3d.py:60-65).  8 (default) = what
 * GPUs report; 4 = the standard's minimum and the software OpenGL tests/golden/gl_raster.npz was drawn with.  4..8. */
int mvlm_set_render_subpixel_bits(mvlm_ctx* ctx, int bits);
/* HIP-event timing of the render kernels (bench.py's rasteriser roofline): while enabled every mvlm_render
 * records one {n_views, n_verts, n_tris, ms}; get_profile waits for the stream, fills up to `cap` records,
 * clears them and returns the count (-1 on failure). */
int mvlm_render_set_profiling(mvlm_ctx* ctx, int enabled);
int mvlm_render_get_profile(mvlm_ctx* ctx, int32_t* n_views, int32_t* n_verts, int32_t* n_tris, float* ms, int cap);
/* mvlm_render only enqueues work; this waits for the stream and reports a deferred failure
 * (tile lists overflowed) of the renders since the last check */
int mvlm_render_check(mvlm_ctx* ctx);

/* ---- landmark network (replaces paulsenpredictor.py:89-110, :167-217) ------------ */
/* blob/desc: output of mvlm_amd.weights.pack_for_device (BN folded, weights as
 * [tap][cin_pad][cout_pad]); copied to the device. */
int mvlm_cnn_load(mvlm_ctx* ctx, const float* blob_host, size_t n_floats, const int32_t* desc_host, int n_slots,
                  int n_landmarks, int in_channels);
/* bytes of caller-provided device scratch needed to push `batch` views at once */
size_t mvlm_cnn_workspace_bytes(mvlm_ctx* ctx, int batch);
/* images_dev f32[N,256,256,4]; chan_sel_host int[in_channels] picks the planes fed to
 * the net; maxima_dev f32[NL,N,3] = (row-1, col-0.5, value) per (landmark, view)
 * (paulsenpredictor.py:123-127).  Views are processed `batch` at a time. */
int mvlm_cnn_maxima(mvlm_ctx* ctx, const float* images_dev, int n_views, const int32_t* chan_sel_host,
                    float* maxima_dev, void* workspace_dev, size_t workspace_bytes, int batch);
/* which maximum mvlm_cnn_maxima reports per (view, landmark) plane (paulsenpredictor.py:112-158): MVLM_MAXIMA_SIMPLE
 * (default) the argmax pixel, MVLM_MAXIMA_MOMENT the 31x31 centroid around it where the peak lies more than 15 pixels
 * from every border.  Both come out of the fused path - the heatmaps are never written: for "moment" the window around
 * each peak is recomputed from conv10's output with conv11's own arithmetic (bit for bit what mvlm_cnn_heatmaps +
 * mvlm_heatmap_maxima(method 1) give).  Fails when the packed weights carry conv11 without its parity slots. */
int mvlm_cnn_set_selection(mvlm_ctx* ctx, int method);
/* same network, but materialises the final-stage heatmaps f32[N,NL,256,256]
 * (what paulsenpredictor.py:187-212 accumulates); used by tests and diagnostics. */
int mvlm_cnn_heatmaps(mvlm_ctx* ctx, const float* images_dev, int n_views, const int32_t* chan_sel_host,
                      float* heat_dev, void* workspace_dev, size_t workspace_bytes, int batch);
/* How the ~140 launches of a forward pass are issued.  graph_mode 1 (default): a pass over the same buffers and
 * shapes is captured as a hipGraph the second time it is seen and replayed afterwards; 0: always launch by
 * launch.  concurrency: 0 (default, the supported mode) = one stream.  1 = EXPERIMENT ONLY: for batches of <= 32 views
 * the lower hourglass pyramid goes to a second stream beside the 128x128 / 64x64 skip blocks - measured 2-10 % SLOWER
 * on the MI355X and kept for that measurement, not for use.  Results are identical in every mode.
 * mvlm_cnn_set_pairing: independent residual blocks of a hourglass level (the skip block and the first block of the
 * next lower level, paulsenpredictor.py:301-361) share launches conv by conv - 0 never, 1 (default) where the measured
 * table (csrc/conv_pair_tuned.h) says one two-problem launch beats two launches, 2 wherever one kernel variant can
 * serve both (tests, tuning).  Per convolution the arithmetic is that of the kernel variant that runs it; a pair may
 * run another variant than the single launch would, so results can differ in the last bits between pairing modes.
 * mvlm_cnn_execution_stats reports how many passes ran eagerly / were captured / replayed, and how many
 * captures failed (those passes ran eagerly instead). */
int mvlm_cnn_set_execution(mvlm_ctx* ctx, int graph_mode, int concurrency);
int mvlm_cnn_set_pairing(mvlm_ctx* ctx, int mode);
int mvlm_cnn_execution_stats(mvlm_ctx* ctx, int64_t* eager_runs, int64_t* graph_captures, int64_t* graph_replays,
                             int64_t* graph_failures);
/* OPT-IN reduced-cost arithmetic ("fast" precision, mvlm_amd/csrc/conv_fast.hip): the big 3x3 layers (input channels a
 * multiple of 16, output channels of 64, 32-pixel rows) multiply bf16x3-split operands on the bf16 matrix cores - 6 of the 9
 * cross products, fp32 accumulation - at 6/16 of the exact path's matrix time.  Results are fp32-accurate but NOT bit-identical
 * to the exact path (argmax near-ties can flip); the default and every parity claim is the exact path.
 *   mvlm_pack_fast_weights: host; w f32[cout][cin][3][3] -> u16 [cin_pad/16][3][3][2][3][cout_pad][8]; returns the element
 *     count (out == NULL: only the count).
 *   mvlm_cnn_load_fast: after mvlm_cnn_load; one blob of packed layers, slot_offsets[n_slots] = u16 offset per conv slot or -1.
 *   mvlm_cnn_set_precision: 0 exact (default), 1 fast (bf16x3), 2 fast16 (f16x2, below).
 *   mvlm_conv2d_fast: test hook like mvlm_conv2d (3x3 only). */
size_t mvlm_pack_fast_weights(const float* w, int cout, int cin, int cout_pad, int cin_pad, uint16_t* out);
int mvlm_cnn_load_fast(mvlm_ctx* ctx, const uint16_t* blob_host, size_t n_u16, const int64_t* slot_offsets, int n_slots);
int mvlm_cnn_set_precision(mvlm_ctx* ctx, int fast);
int mvlm_conv2d_fast(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                     const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                     const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev);
/* Second opt-in form, mvlm_cnn_set_precision(ctx, 2) ("fast16"): the same layers on f16x2-split operands - x = h + l as two
 * fp16 terms (22 significant bits), the three cross products xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation: 3/16 of the exact path's matrix time.  Per product 3-4 x the error of the bf16x3 form, still of the order of
 * an fp32 summation's own rounding.  The weights are scaled per layer by a power of two on the host (fp16's exponent is
 * narrow; the epilogue multiplies by the inverse, exactly); an activation with |x| >= 65504 has no fp16 form: the kernel
 * raises a flag, the pass's maxima scores (mvlm_cnn_maxima) / the first value of every heatmap plane (mvlm_cnn_heatmaps)
 * come back as NaN, and the Python layer repeats the pass with the bf16x3 form.
 *   mvlm_pack_fast_weights16: like mvlm_pack_fast_weights with two splits per operand; *unscale = the inverse scale.
 *   mvlm_cnn_load_fast16: after mvlm_cnn_load; slot_unscale[n_slots] = the inverse scale per conv slot.
 *   mvlm_cnn_fast16_overflowed: *overflowed = 1 when the last mvlm_cnn_maxima / mvlm_cnn_heatmaps call on this context met
 *     such an activation (waits for the context's stream).
 *   mvlm_conv2d_fast16: test hook like mvlm_conv2d_fast. */
size_t mvlm_pack_fast_weights16(const float* w, int cout, int cin, int cout_pad, int cin_pad, uint16_t* out, float* unscale);
int mvlm_cnn_load_fast16(mvlm_ctx* ctx, const uint16_t* blob_host, size_t n_u16, const int64_t* slot_offsets,
                         const float* slot_unscale, int n_slots);
int mvlm_cnn_fast16_overflowed(mvlm_ctx* ctx, int* overflowed);
int mvlm_conv2d_fast16(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                       const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                       const float* post_scale_host, const float* post_shift_host, const float* r_dev, float* y_dev);
/* per-kernel timing of the last mvlm_cnn_* call when profiling is on: fills up to
 * `cap` records of {slot, kernel_variant, flops, ms}; returns the record count. */
int mvlm_cnn_set_profiling(mvlm_ctx* ctx, int enabled);
int mvlm_cnn_get_profile(mvlm_ctx* ctx, int32_t* slot, int32_t* variant, double* flops, float* ms, int cap);
const char* mvlm_conv_variant_name(int variant);
/* shapes of those records, in the same order: {ksize, cin_pad, cout_pad, size, kind, batch} per record (kind: 0 plain /
 * residual-block layer, 1 scatter into the skip tensor, 2 pooled output wanted; slot -1 = the pool kernel: channels in
 * cout_pad).  Returns the record count. */
int mvlm_cnn_get_profile_shapes(mvlm_ctx* ctx, int32_t* shapes6, int cap);
/* tuning hooks of tools/tune_in_network.py: mvlm_conv_variant_serves - can kernel variant `variant` run a 3x3 layer of this
 * shape and kind (1 / 0); mvlm_conv_set_override - every launch of this (shape, kind) on this context runs that variant,
 * ahead of the measured tables (variant < 0 removes the entry, ksize 0 all of them). */
int mvlm_conv_variant_serves(int variant, int ksize, int cin_pad, int cout_pad, int size, int kind);
int mvlm_conv_set_override(mvlm_ctx* ctx, int ksize, int cin_pad, int cout_pad, int size, int kind, int variant);

/* heatmap maxima of materialised heatmaps (replaces paulsenpredictor.py:112-165).
 * heat_dev f32[N,NL,S,S] -> out_dev f32[NL,N,3]. */
int mvlm_heatmap_maxima(mvlm_ctx* ctx, const float* heat_dev, int n_views, int n_landmarks, int size, int method,
                        float* out_dev);

/* one generic convolution launch (the kernel the network is made of); test hook.
 * x_dev f32[B,Cin,H,W] -> y_dev f32[B,Cout,H,W]; w_host f32[Cout,Cin,k,k] (k = 1|3,
 * stride 1, pad k/2); optional bias / pre-BN+ReLU (scale,shift per Cin) / post-BN+ReLU
 * (per Cout) / residual add r_dev f32[B,Cout,H,W]. */
int mvlm_conv2d(mvlm_ctx* ctx, const float* x_dev, int batch, int cin, int h, int w, const float* w_host, int cout,
                int ksize, const float* bias_host, const float* pre_scale_host, const float* pre_shift_host,
                const float* post_scale_host, const float* post_shift_host, const float* r_dev, int upsample_in,
                float* y_dev);

/* test hook for the two-problem launch: x_i f32[B,cin,size_i,size_i] * w_i f32[cout,cin,3,3] (shared pre-BN+ReLU per
 * input channel, optional residual r_i and raw copy raw_i, both f32[B,cout,size_i,size_i]) -> y_i, both in ONE grid of
 * kernel variant (variant & 255) with 1 << ((variant >> 8) & 3) / 1 << ((variant >> 10) & 3) K parts.  The results equal
 * mvlm_conv2d's with that variant forced, bit for bit. */
int mvlm_conv2d_pair(mvlm_ctx* ctx, int batch, int cin, int cout, const float* x0_dev, int size0, const float* w0_host,
                     const float* r0_dev, float* raw0_dev, float* y0_dev, const float* x1_dev, int size1,
                     const float* w1_host, const float* r1_dev, float* raw1_dev, float* y1_dev,
                     const float* pre_scale_host, const float* pre_shift_host, int variant);
/* test hook: mvlm_conv2d on this ctx runs kernel variant `variant` (ids of mvlm_conv_variant_name; -1 = automatic) */
int mvlm_conv_force_variant(mvlm_ctx* ctx, int variant);
/* kernel-variant timing for tools/tune_conv.py: `iters` launches of one layer shape on zero data with kernel
 * variant `variant` (< 0: the dispatcher's choice, reported in *variant_used); flags: 1 pre-BN+ReLU, 2 residual
 * add + raw copy, 4 bias, 8 post-BN+ReLU.  Fails for shapes the variant cannot serve. */
int mvlm_conv_bench(mvlm_ctx* ctx, int batch, int cin, int cout, int ksize, int size, int flags, int variant, int iters,
                    float* ms_per_launch, int* variant_used);
/* the same for tools/tune_conv_pairs.py: the 3x3 layer at `size` and at size / 2 (conv j of two independent residual
 * blocks) as the two tuned single launches (variant < 0) or as ONE two-problem launch of kernel variant (variant & 255)
 * with 1 << ((variant >> 8) & 3) and 1 << ((variant >> 10) & 3) K parts for the two problems. */
int mvlm_conv_pair_bench(mvlm_ctx* ctx, int batch, int cin, int cout, int size, int flags, int variant, int iters,
                         float* ms_per_launch);

/* ---- rays + consensus (replaces estimator3d.py:31-90, :92-183, utils3d.py:99-124) - */
/* maxima_dev f32[NL,N,3], rot_dev f64[N,9] -> starts_dev, ends_dev f64[NL,N,3] */
int mvlm_estimate_lines(mvlm_ctx* ctx, const float* maxima_dev, const double* rot_dev, int n_views, int n_landmarks,
                        int image_size, double* starts_dev, double* ends_dev);
/* view filter (estimator3d.py:140-155): mask_dev u8[NL,N], count_dev i32[NL] */
int mvlm_consensus_mask(mvlm_ctx* ctx, const float* maxima_dev, int n_views, int n_landmarks, int mode, double q,
                        double thr, uint8_t* mask_dev, int32_t* count_dev);
/* one wavefront per landmark: one-shot RANSAC + least squares (estimator3d.py:92-137,
 * :173-181).  draws_dev i32[NL,8] = the host's np.random.choice draws (indices into the
 * masked line list; ignored where count < 3).  out_dev f64[NL,3], err_dev f64[NL]
 * (the landmark's contribution to sum_error; 0 where count < 3). */
int mvlm_consensus_solve(mvlm_ctx* ctx, const double* starts_dev, const double* ends_dev, const uint8_t* mask_dev,
                         const int32_t* count_dev, const int32_t* draws_dev, int n_views, int n_landmarks,
                         double* out_dev, double* err_dev);

/* ---- surface snap (replaces estimator3d.py:252-285) ------------------------------ */
/* pts_dev f64[NL,3] -> out_dev f64[NL,3]: closest point on the triangle surface */
int mvlm_project_to_surface(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* pts_dev, int n_points,
                            double* out_dev);

/* ---- ray clipping / depth unprojection (replaces visualization/ray_visualizer.py:172-192) */
/* starts_dev, ends_dev f64[R,3] (e.g. the [NL,N,3] arrays of mvlm_estimate_lines) -> new_ends_dev f64[R,3]:
 * the first intersection of each segment with the triangle surface, or the old end when it misses;
 * hit_dev u8[R] (may be NULL) flags the hits.  Along a view ray this is the surface point the view's
 * depth buffer stores, evaluated at the maximum's sub-pixel position. */
int mvlm_clip_rays_to_mesh(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* starts_dev, const double* ends_dev,
                           int n_rays, double* new_ends_dev, uint8_t* hit_dev);

/* ---- multi-GPU: the one exchange of the sharded path (the reference's counterpart is the dormant
 * nn.DataParallel of paulsenpredictor.py:100-105) ---------------------------------------------------------------------
 * Views are sharded contiguously over `world` ranks (rank r holds views [r*base + min(r, rem), ...), sizes differing by at most
 * one: mvlm_amd/parallel.py shard_range); every rank passes the maxima of ITS views, f32[NL, n_local, 3] (NULL when it holds
 * none), and receives all of them in pose-table order, f32[NL, n_total, 3], on the context's stream: pack, ONE ncclAllGather
 * over `nccl_comm` (an ncclComm_t of the caller's RCCL, whose device is this context's), unpack.  RCCL is looked up at first
 * use (MVLM_RCCL_LIB = path, else the RCCL already loaded in the process, else librccl.so.1): nothing links against it.
 * nccl_comm == NULL is allowed in a world of one.  The Python host uses torch.distributed instead (same layout). */
int mvlm_allgather_maxima(mvlm_ctx* ctx, void* nccl_comm, int rank, int world, const float* maxima_local_dev, int n_total,
                          int n_landmarks, float* maxima_all_dev);
/* The two halves around the transport, for a host that moves the slots itself (MPI, hipMemcpyPeerAsync): a rank's slot is
 * f32[n_max, NL, 3], n_max = ceil(n_total / world), short shards zero-padded; slots_dev = the `world` slots in rank order. */
int mvlm_gather_pack(mvlm_ctx* ctx, const float* maxima_local_dev, int n_local, int n_max, int n_landmarks, float* slot_dev);
int mvlm_gather_unpack(mvlm_ctx* ctx, const float* slots_dev, int world, int n_total, int n_landmarks, float* maxima_all_dev);

#ifdef __cplusplus
}
#endif
#endif /* MVLM_HIP_H */
