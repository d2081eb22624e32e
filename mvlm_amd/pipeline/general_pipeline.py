"""``Pipeline.predict_one_file`` - the hot path's orchestrator.

Same class surface as the reference's ``Pipeline``
(src/mvlm/pipeline/general_pipeline.py:39-146): constructor kwargs, the three
duck-typed slots ``renderer_3d`` / ``predictor_2d`` / ``estimator_3d``,
``get_lm_count`` and ``predict_one_file(file_name, landmark_indices=None,
view_indices=None, clip_rays_to_mesh=True) -> ndarray[NL,3] | None``.

When all three slots are this package's HIP objects the call runs *fused*: the
rendered views, the maxima and the rays never leave HBM; only the per-landmark
survivor counts (for the host-side RANSAC draw) and the final [NL,3] result cross
PCIe.  Any other object in a slot (a user predictor, the reference's own classes)
is driven through the reference's numpy slot protocol instead.
"""
from __future__ import annotations

import abc
import functools
import os
import threading
import time
from pathlib import Path

import numpy as np

from .. import parallel
from ..prediction.paulsenpredictor import HipPaulsenModel
from ..utils.estimator3d import HipEstimator3D
from ..utils.hostmem import retain_freed_host_memory
from ..utils.render3d import HipRenderer3D

__all__ = ["Pipeline"]


def _serialised(method):
    """One caller at a time per pipeline.  The reference's server calls ``predict_one_file`` from a thread pool
    without locks (3DMD_server.py:26-31); here a call owns the pipeline's device buffers (image stack, maxima,
    result pack), its captured launch graphs, ``timings`` and the per-call ray state, so the whole call - RNG
    draws included - runs under the pipeline's re-entrant lock."""
    @functools.wraps(method)
    def locked(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)
    return locked


def _drop(box: list) -> None:
    """Release the last reference to whatever ``box`` holds on the calling (reader) thread."""
    box.clear()


class StageTimer:
    """Wall-clock seconds per stage of the last call, under the reference's stage names
    (the reference prints them through tic/toc, general_pipeline.py:84-111; here they are kept
    as a dict so callers and bench.py can read them, and printed in the same wording when verbose)."""

    LABELS = {"render": "Render [Total]: ", "prediction": "Prediction [Total]: ",
              "lines": "Landmarks [0] - From Heatmaps: ", "consensus": "Landmarks [1] - From View Lines: ",
              "project": "Landmarks [2] - Project to Surface: ", "total": "Landmarks 3D Total: "}

    def __init__(self, sink: dict, say):
        self.sink, self.say = sink, say

    @staticmethod
    def fmt(seconds: float) -> str:
        return f"{seconds:08.6f} s"

    def stage(self, name: str):
        return _Stage(self, name)


class _Stage:
    def __init__(self, timer: StageTimer, name: str):
        self.timer, self.name = timer, name

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        dt = time.perf_counter() - self.t0
        self.timer.sink[self.name] = dt
        if exc[0] is None:
            self.timer.say(StageTimer.LABELS.get(self.name, self.name + ": "), StageTimer.fmt(dt))
        return False


class Pipeline(abc.ABC):
    def __init__(self, render_image_stack: bool = False, offscreen: bool = True, n_views: int = 8,
                 render_image_folder: Path | None = None, visualize_rays: bool = False,
                 screenshot_folder: Path | None = None, device: int = 0, shard_views: bool = False,
                 verbose: bool = True):
        self.render_image_stack = render_image_stack
        self.render_image_folder = render_image_folder
        self.n_views = n_views
        self.visualize_rays = visualize_rays
        self.screenshot_folder = screenshot_folder
        self.device = device
        self.shard_views = shard_views
        self.verbose = verbose
        self.timings: dict[str, float] = {}
        self._timer = StageTimer(self.timings, self._say)
        self.last_error: float | None = None
        self._rays = None  # (mesh, starts, ends) of the current call when visualize_rays is set
        self._buffers: dict = {}
        self._lock = threading.RLock()  # see _serialised
        self._pre_align: dict | None = None
        self.write_pre_aligned_folder: Path | None = None

        retain_freed_host_memory()  # scans come and go: keep their host pages (utils/hostmem.py; MVLM_HOST_MALLOC_TUNING=0 opts out)
        if shard_views:
            # one process per GPU: collectives (RCCL) and the allocator must use THIS rank's device, not cuda:0
            import torch

            if torch.cuda.is_available():
                torch.cuda.set_device(device)
        self.renderer_3d = HipRenderer3D(image_size=(256, 256), offscreen=offscreen, n_views=n_views, device=device,
                                         verbose=verbose)
        self.estimator_3d = HipEstimator3D(device=device, verbose=verbose)
        self.predictor_2d = None  # will be assigned externally

    @property
    def pre_align(self) -> dict | None:
        """Optional "pre-align" block of a Deep-MVLM config (mvlm_amd/utils/prealign.py, utils3d.py:465-527); the
        reference's live pipeline has none (it renders the mesh as-is).  It lives on the renderer slot, which is
        the object that loads meshes; both the fused path and the slot protocol honour it."""
        return getattr(self.renderer_3d, "pre_align", self._pre_align)

    @pre_align.setter
    def pre_align(self, block: dict | None) -> None:
        from ..utils.prealign import is_active

        self._pre_align = block
        if hasattr(self.renderer_3d, "pre_align"):
            self.renderer_3d.pre_align = block
        elif is_active(block):
            raise ValueError("a pre-align block needs a renderer that applies it (HipRenderer3D.pre_align)")

    def _texture_needed(self) -> bool:
        """Does anything downstream read a texture-shaded plane of the views?  The reference always decodes the JPEG
        (utils3d.py:26-36) and renders RGB + depth; a depth or geometry(+depth) model then never looks at the colours.
        Kept: whenever the views are written out (render_image_stack) or go to a predictor whose planes are unknown."""
        sel = getattr(self.predictor_2d, "chan_sel", None)
        if self.render_image_stack or sel is None or not isinstance(self.renderer_3d, HipRenderer3D):
            return True
        if self.renderer_3d.shading == "geometry":
            return False  # planes 0..2 carry the build-defined geometry shading
        return any(int(c) < 3 for c in sel)

    def _to_original(self, mesh, landmarks):
        """Landmarks found on a pre-aligned mesh -> the file's own coordinates (utils3d.py:505-527)."""
        m = getattr(mesh, "to_original", None)
        if m is None or landmarks is None:
            return landmarks
        from ..utils.prealign import landmarks_to_original_space

        return landmarks_to_original_space(landmarks, m)

    def get_lm_count(self) -> int:
        if self.predictor_2d is None:
            raise ValueError("Predictor2D is not initialized.")
        return self.predictor_2d.get_lm_count()

    def _buffer(self, name: str, shape: tuple):
        """A float32 device tensor of this shape that lives as long as the pipeline (one per name)."""
        import torch

        buf = self._buffers.get(name)
        if buf is None or tuple(buf.shape) != tuple(shape):
            buf = torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", self.device))
            self._buffers[name] = buf
        return buf

    def _pinned_bytes(self, name: str, n_bytes: int):
        """(pinned uint8 host tensor of exactly this size, its numpy view) that live as long as the pipeline."""
        import torch

        ent = self._buffers.get(name)
        if ent is None or ent[0].numel() != n_bytes:
            t = torch.empty(n_bytes, dtype=torch.uint8).pin_memory()
            ent = self._buffers[name] = (t, t.numpy())
        return ent

    def _buffer_bytes(self, name: str, n_bytes: int):
        """A uint8 device buffer of at least this size that lives as long as the pipeline (8-byte aligned slices)."""
        import torch

        buf = self._buffers.get(name)
        if buf is None or buf.numel() < n_bytes:
            buf = torch.empty((n_bytes + 7) // 8 * 8, dtype=torch.uint8, device=torch.device("cuda", self.device))
            self._buffers[name] = buf
        return buf

    def _say(self, *a):
        if self.verbose:
            print(*a)

    def _fusable(self) -> bool:
        # any predictor that can hand over device-resident maxima for every view (HipPaulsenModel; a
        # PrecomputedPredictor built with ``device_fn``) keeps the whole call in HBM
        p2 = self.predictor_2d
        device_predictor = isinstance(p2, HipPaulsenModel) or callable(getattr(p2, "predict_device", None))
        return (isinstance(self.renderer_3d, HipRenderer3D) and isinstance(self.estimator_3d, HipEstimator3D)
                and device_predictor)

    def predict_one_file(self, file_name: Path, landmark_indices: list[int] | None = None,
                         view_indices: list[int] | None = None, clip_rays_to_mesh: bool = True):
        if self.predictor_2d is None:
            raise ValueError("Predictor2D is not initialized.")
        file_name = Path(file_name)
        if not file_name.exists():
            print(f"File {file_name} does not exist")
            return None
        # File ingest (OBJ parse, JPEG read - or decode, with texture_decode="host" -: host work, 3-20 ms per scan) and the
        # upload (pinned staging, the library's own copy stream, the JPEG's decode on the device) run BEFORE the pipeline
        # lock is taken: the callers of a server's thread pool (3DMD_server.py:26-31) then ingest their scans side by side
        # and only the GPU section - RNG draws, device buffers, launch graphs, timings - is one caller at a time.
        mesh, load_seconds = None, 0.0
        if self._fusable():
            from ..utils.render3d import upload_mesh

            t0 = time.perf_counter()
            mesh = self.renderer_3d.load_mesh(self.renderer_3d._check_file(file_name), load_texture=self._texture_needed())
            if mesh.n_tris > 0:
                upload_mesh(self.renderer_3d.ctx, mesh)  # (thread-safe beside a launching thread: include/mvlm_hip.h)
            load_seconds = time.perf_counter() - t0
        with self._lock:
            self._rays = None
            with self._timer.stage("total"):
                if mesh is not None:
                    landmarks = self._predict_fused(file_name, mesh=mesh)
                else:
                    landmarks = self._predict_slots(file_name)
            if mesh is not None:
                self.timings["load"] = self.timings.get("load", 0.0) + load_seconds
                self.timings["total"] += load_seconds
            self._after_prediction(file_name, landmarks, landmark_indices, view_indices, clip_rays_to_mesh)
        return landmarks

    def _after_prediction(self, file_name, landmarks, landmark_indices=None, view_indices=None, clip_rays_to_mesh=True):
        """What follows every prediction, whichever entry point made it (predict_one_file, predict_files):
        the optional ray dump (general_pipeline.py:111-130) and the reset of the per-call ray state."""
        if self.visualize_rays and self._rays is not None:
            try:
                self.dump_rays(file_name, landmarks, landmark_indices, view_indices, clip_rays_to_mesh)
            except Exception as e:  # noqa: BLE001 - general_pipeline.py:129-130: a failed visualisation never fails the call
                print(f"[Pipeline] Ray visualization failed: {e}")
        self._rays = None

    def dump_rays(self, file_name: Path, landmarks, landmark_indices=None, view_indices=None, clip_to_mesh: bool = True):
        """What the reference hands to its VTK ``RayVisualizer`` (general_pipeline.py:111-128), written
        as ``<stem>_rays.npz`` for any viewer: the selected view rays (``starts``, ``ends`` [L,V,3]),
        clipped to their first hit with the mesh when ``clip_to_mesh`` (ray_visualizer.py:172-192,
        here on the GPU: mvlm_clip_rays_to_mesh), ``hit`` [L,V], the indices and the landmarks."""
        mesh, starts, ends = self._rays
        lm_idx = list(range(starts.shape[0])) if landmark_indices is None else [int(i) % starts.shape[0] for i in landmark_indices]
        v_idx = list(range(starts.shape[1])) if view_indices is None else [int(i) % starts.shape[1] for i in view_indices]
        fs = np.ascontiguousarray(starts[np.ix_(lm_idx, v_idx)])
        fe = np.ascontiguousarray(ends[np.ix_(lm_idx, v_idx)])
        hit = np.zeros(fs.shape[:2], bool)
        if clip_to_mesh and fs.size:
            fe, hit = self.estimator_3d.clip_rays_to_mesh(mesh, fs, fe)
        folder = Path(self.screenshot_folder) if self.screenshot_folder else Path.cwd() / "visualization" / "ray_screenshots"
        folder.mkdir(parents=True, exist_ok=True)
        out = folder / f"{Path(file_name).stem}_rays.npz"
        np.savez(out, starts=fs, ends=fe, hit=hit, landmark_indices=np.asarray(lm_idx), view_indices=np.asarray(v_idx),
                 landmarks=np.asarray(landmarks), clipped=bool(clip_to_mesh))
        self._say(f"[Pipeline] rays written to {out}")
        return out

    # ---- fused device-resident path ----------------------------------------------------
    @_serialised
    def predict_mesh_device(self, mesh, transform_stack):
        """Render + network + fusion + snap for an already loaded mesh and pose table.
        Returns (landmarks [NL,3] float64 numpy, mean RANSAC error).  With
        ``shard_views`` under torch.distributed each rank handles a slice of the views
        (a rank whose slice is empty - more ranks than views - skips render and network
        but still joins the collectives)."""
        import torch

        r3, p2, e3 = self.renderer_3d, self.predictor_2d, self.estimator_3d
        n_total = int(transform_stack.shape[0])
        sharded = self.shard_views and parallel.is_distributed()
        rank, world = parallel.rank_world() if sharded else (0, 1)
        lo, hi = parallel.shard_range(n_total, rank, world)

        from ..utils.render3d import view_rotations

        rot = view_rotations(transform_stack)  # once per call: renderer and estimator share it
        tm = self._timer
        dev_t = torch.device("cuda", self.device)
        # this call owns the renderer's / estimator's context until it returns: the launch stream is bound once
        with r3.ctx.hold_stream(torch, dev_t), e3.ctx.hold_stream(torch, dev_t):
            return self._predict_mesh_device_held(mesh, transform_stack, rot, r3, p2, e3, tm, dev_t, n_total, sharded, rank, lo, hi)

    def _predict_mesh_device_held(self, mesh, transform_stack, rot, r3, p2, e3, tm, dev_t, n_total, sharded, rank, lo, hi):
        import torch

        # What the GPU waits for first is the render: single-process, it is enqueued before anything else the host has to
        # prepare (the rotations' second upload, the result buffers, the RANSAC draws: all of that then happens while the
        # rasteriser runs - at 0.5 ms per step, configs[4], the host's serial part is what the step time is made of).
        # Sharded, the draws come out of a collective, which must not queue up behind this step's network: they go first.
        draws_fn = None
        if sharded:
            # one RNG stream for the job: rank 0 draws (global numpy RNG, as the reference) and broadcasts the table
            def draws_fn(counts, keep_on_device=False):
                draws = e3.draw_ransac_indices(counts) if rank == 0 else None
                return parallel.broadcast_int32(draws, (len(counts), 8), dev_t, keep_on_device=keep_on_device)

            draws_fn.device_result = True  # plan_draws may ask for the collective's device tensor (RCCL)

        nl_all = p2.get_lm_count()
        plan = e3.plan_draws(nl_all, n_total, draws_fn) if sharded else None
        with tm.stage("render"):
            # one image stack / maxima buffer per view count, reused from call to call: stable addresses let the
            # predictor replay its captured launch graph instead of re-enqueueing ~160 kernels per mesh
            images = None
            if hi > lo:
                images = r3.render_device(mesh, transform_stack[lo:hi], rot=rot[lo:hi],
                                          out=self._buffer("images", (hi - lo, 256, 256, 4)))
            if self.verbose:
                torch.cuda.synchronize()
        if images is not None and hi - lo == n_total and r3.ctx is e3.ctx:
            rot_dev = r3.rotations_device()   # all views rendered here: the rays read the rasteriser's own device copy
        else:
            # (enqueued behind the render on the same stream, through pinned staging: the host does not wait)
            rot_dev = e3.upload_rotations_async(rot)
        # landmarks f64[NL,3] | error f64[NL] | survivor counts i32[NL] in ONE buffer: one device-to-host copy per mesh,
        # into pinned memory, enqueued right behind the last kernel
        pack = self._buffer_bytes("result", nl_all * (24 + 8 + 4))
        pack_host, pack_np = self._pinned_bytes("result_host", int(pack.numel()))
        snap_view = pack[: nl_all * 24].view(torch.float64).view(nl_all, 3)
        err_view = pack[nl_all * 24: nl_all * 32].view(torch.float64)
        count_view = pack[nl_all * 32: nl_all * 36].view(torch.int32)
        if self.render_image_stack and images is not None:
            self.visualize_image_stack(images.cpu().numpy(), mesh.path or Path("mesh.obj"), first_index=lo)

        valid = None  # host bool [views of this rank]: a detector's "nothing found in this view" (None: all valid)
        with tm.stage("prediction"):
            if images is not None and isinstance(p2, HipPaulsenModel):
                maxima = p2.predict_device(images, out=self._buffer("maxima", (p2.get_lm_count(), hi - lo, 3)))
            elif images is not None:
                maxima = p2.predict_device(images)
                if isinstance(maxima, tuple):
                    maxima, valid = maxima
            else:
                maxima = torch.empty((p2.get_lm_count(), 0, 3), dtype=torch.float32,
                                     device=torch.device("cuda", self.device))
            del images
            if sharded:
                maxima = parallel.all_gather_views(maxima, n_total)
                if not isinstance(p2, HipPaulsenModel):  # (same predictor class on every rank: all join or none does)
                    valid = parallel.all_gather_valid(valid, hi - lo, n_total, self.device)
            if self.verbose:
                torch.cuda.synchronize()
        if valid is not None and not valid.all():
            # views without a detection leave the call here, as in the reference (general_pipeline.py:93-95:
            # landmark_stack[:, valid], transform_stack[valid]) - before rays, filter and the RANSAC draws, which
            # address the remaining views by position
            keep = np.nonzero(valid)[0]
            maxima = maxima.index_select(1, torch.from_numpy(keep).to(maxima.device)).contiguous()
            transform_stack = transform_stack[keep]
            rot_dev = e3.upload_rotations_async(rot[keep])
            n_total = int(len(keep))
            if plan is not None:  # sharded: the draws planned for all views are void - plan again for the views that remain
                np.random.set_state(plan["rng_state"])
                plan = e3.plan_draws(nl_all, n_total, draws_fn)

        if plan is None:
            # the RANSAC draws for the expected survivor counts (~0.3 ms of numpy calls) are made now, while the GPU
            # works on the views; they travel through pinned memory on a copy stream of their own
            plan = e3.plan_draws(nl_all, n_total, None)

        with tm.stage("lines"):
            starts, ends = e3.lines_device(maxima, transform_stack, 256, rot_dev=rot_dev)

        with tm.stage("consensus"):
            if self.visualize_rays:
                self._rays = (mesh, starts.cpu().numpy(), ends.cpu().numpy())
            out, err, verify = e3.consensus_device(maxima, starts, ends, deferred=True, plan=plan, err_out=err_view,
                                                   count_out=count_view)

        with tm.stage("project"):
            # the snap is enqueued before anything is fetched: one wait and one copy per mesh, at the end
            e3.project_device(mesh, out, out=snap_view)
            stream = torch.cuda.current_stream(dev_t)
            pack_host.copy_(pack, non_blocking=True)
            stream.synchronize()
            host = pack_np
            if verify(counts=host[nl_all * 32: nl_all * 36].view(np.int32)):
                # the RANSAC draws had to be repeated for other survivor counts (tied / NaN scores, absolute mode)
                e3.project_device(mesh, out, out=snap_view)
                pack_host.copy_(pack, non_blocking=True)
                stream.synchronize()
            landmarks = host[: nl_all * 24].view(np.float64).reshape(nl_all, 3).copy()
            error = e3.mean_error(host[nl_all * 24: nl_all * 32].view(np.float64))
            r3.check()  # deferred renderer status (the stream has been waited for above)
        asked16 = getattr(p2, "configured_precision", getattr(p2, "precision", None)) == "fast16"
        if asked16:
            # an activation beyond fp16's range somewhere in the network (the kernel raised the context's flag; the maxima of
            # the pass carry NaN scores, which survive no filter - the landmarks above are finite nonsense).  Same scan again
            # on bf16x3; asked here, after the step's own wait for its results, the question costs one 4-byte copy.
            # (the global RNG stands behind this call's draws: the repeat draws again, as a second call would.)
            # Sharded: the decision is taken by all ranks together - a rank repeating alone would wait in collectives nobody
            # else joins - and WHETHER to ask is decided from what the caller configured, which is the same on every rank:
            # a rank that already fell back on its own (predict_landmarks_from_images) still joins the all-reduce.
            mine = p2.precision == "fast16" and p2.fast16_overflowed()
            if parallel.any_rank(mine, self.device) if sharded else mine:
                return p2.repeat_without_fp16(lambda: self.predict_mesh_device(mesh, transform_stack))
        self._say("Landmarks [Error]: ", f"{error:08.6f}", " mm")
        self.last_error = error
        return landmarks, error

    def _groupable(self, n_scans: int) -> bool:
        """Can ``n_scans`` scans share one network pass (predict_meshes_device)?"""
        p2, e3 = self.predictor_2d, self.estimator_3d
        if n_scans < 2 or not self._fusable() or not isinstance(p2, HipPaulsenModel):
            return False
        if p2.precision == "fast16":
            return False  # the fp16 range guard (and its fallback) works scan by scan
        if (self.shard_views and parallel.is_distributed()) or self.render_image_stack or self.visualize_rays:
            return False
        n = int(self.renderer_3d.n_views)
        return e3.expected_counts(p2.get_lm_count(), n) is not None and n_scans * n <= (p2.device_batch or 128)

    @_serialised
    def predict_meshes_device(self, meshes, pose_fn=None):
        """Several loaded scans through ONE pass of the network: with few views per scan (the reference's default 8) a
        pass over 8 views runs the matrix cores at 0.64 of what a pass over 64 does.  Every scan is rendered into its
        slice of one image stack, the network runs once, rays / consensus / snap follow per scan and ONE copy brings all
        results back.  The global RNG is consumed exactly as by a loop of ``predict_mesh_device``: poses of scan 1, draws
        of scan 1, poses of scan 2, ... - the draws being the speculative ones of ``plan_draws``; if a scan's survivor
        counts differ from the expectation (tied / NaN scores), that scan's draws are repeated from the saved RNG
        state and the scans after it are processed one by one from there.  Returns [(landmarks, error), ...]; falls back
        to the loop when the scans cannot share a pass (``_groupable``)."""
        import torch

        from ..utils.render3d import view_rotations

        r3, p2, e3 = self.renderer_3d, self.predictor_2d, self.estimator_3d
        pose_fn = pose_fn or r3.generate_3d_transformations
        k = len(meshes)
        from ..utils.prealign import aligned

        meshes = [aligned(m, self.pre_align) for m in meshes]
        if not self._groupable(k):
            out = []
            for m in meshes:
                landmarks, err = self.predict_mesh_device(m, pose_fn())
                out.append((self._to_original(m, landmarks), err))
            return out
        nl, n = p2.get_lm_count(), int(r3.n_views)
        per = (nl * 36 + 7) // 8 * 8
        stacks, rots, rot_devs, plans = [], [], [], []
        for j in range(k):  # the host's RNG work first, in the loop's order; uploads while the stream is empty
            ts = pose_fn()
            if int(ts.shape[0]) != n:
                raise ValueError("predict_meshes_device: every scan needs the pipeline's n_views poses")
            rot = view_rotations(ts)
            stacks.append(ts)
            rots.append(rot)
            plans.append(e3.plan_draws(nl, n, None, slot=j))
            rot_devs.append(e3.upload_rotations(rot))
        tm = self._timer
        with tm.stage("render"):
            images = self._buffer("images_group", (k * n, 256, 256, 4))
            for j in range(k):
                r3.render_device(meshes[j], stacks[j], rot=rots[j], out=images[j * n:(j + 1) * n])
        with tm.stage("prediction"):
            maxima = p2.predict_device(images, out=self._buffer("maxima_group", (nl, k * n, 3)))
        pack = self._buffer_bytes("result_group", k * per)
        pending = []
        with tm.stage("consensus"):
            for j in range(k):
                seg = pack[j * per:(j + 1) * per]
                snap_view = seg[: nl * 24].view(torch.float64).view(nl, 3)
                err_view = seg[nl * 24: nl * 32].view(torch.float64)
                count_view = seg[nl * 32: nl * 36].view(torch.int32)
                mj = maxima[:, j * n:(j + 1) * n].contiguous()
                starts, ends = e3.lines_device(mj, stacks[j], 256, rot_dev=rot_devs[j])
                out, _, verify = e3.consensus_device(mj, starts, ends, deferred=True, plan=plans[j], err_out=err_view,
                                                     count_out=count_view)
                e3.project_device(meshes[j], out, out=snap_view)
                pending.append((out, verify, snap_view))
        results = []

        def unpack(host_seg):
            landmarks = host_seg[: nl * 24].view(np.float64).reshape(nl, 3).copy()
            return landmarks, e3.mean_error(host_seg[nl * 24: nl * 32].view(np.float64))

        def original(j, res):
            return self._to_original(meshes[j], res[0]), res[1]

        with tm.stage("project"):
            host = pack[: k * per].cpu().numpy()
            for j in range(k):
                out, verify, snap_view = pending[j]
                seg = host[j * per:(j + 1) * per]
                if verify(counts=seg[nl * 32: nl * 36].view(np.int32)):
                    # other survivor counts than planned: this scan's draws + solve were repeated from its saved RNG
                    # state; the RNG now stands where the loop would have it, so the remaining scans go one by one
                    e3.project_device(meshes[j], out, out=snap_view)
                    results.append(original(j, unpack(pack[j * per:(j + 1) * per].cpu().numpy())))
                    for i in range(j + 1, k):
                        results.append(original(i, self.predict_mesh_device(meshes[i], pose_fn())))
                    break
                results.append(original(j, unpack(seg)))
            r3.check()
        self.last_error = results[-1][1]
        return results

    def predict_files(self, files, prefetch: int = 2, readers: int | None = None, batch_scans: int = 1):
        """``predict_one_file`` over many scans, yielding ``(file, landmarks | None)`` in order.

        The reference's CLI loops ``predict_one_file`` (main.py:55-62), paying file ingest and GPU
        work back to back.  Here a reader thread parses the next ``prefetch`` OBJ/JPEG pairs
        (native reader + libjpeg, both outside the GIL) while the GPU works on the current scan,
        so a folder runs at the GPU rate.  With few views per scan (the reference's default 8) one
        reader is slower than the GPU: ``readers`` threads (default: a quarter of the host's cores,
        1..4) parse ``max(prefetch, readers + 1)`` scans ahead; delivery stays in file order.  Results
        equal the sequential loop's: poses and RANSAC draws are taken on the calling thread in file
        order, and every scan gets the same post-step (ray dump) as ``predict_one_file``.
        ``batch_scans`` > 1: that many consecutive scans share one pass of the network
        (``predict_meshes_device``; same results, higher throughput when a scan has few views)."""
        from concurrent.futures import ThreadPoolExecutor

        from ..utils.mesh_io import load_obj
        from ..utils.render3d import upload_mesh

        files = [Path(f) for f in files]
        if self.predictor_2d is None:
            raise ValueError("Predictor2D is not initialized.")
        if not self._fusable() or prefetch <= 0:
            for f in files:
                yield f, self.predict_one_file(f)
            return

        load_texture = self._texture_needed()

        def ingest(f: Path):
            if not f.exists():
                return None
            mesh = self.renderer_3d.load_mesh(self.renderer_3d._check_file(f), load_texture=load_texture)  # pre-aligned here, uploaded once
            if mesh.n_tris > 0:
                # device copy from the reader thread too: pinned staging + a copy stream of the library's own, so the
                # transfer runs beside the current scan's kernels (mvlm_mesh_upload); the renderer waits for its event
                upload_mesh(self.renderer_3d.ctx, mesh)
            return mesh

        if readers is None:
            readers = min(4, max(1, (os.cpu_count() or 4) // 4))
        readers = max(1, int(readers))
        batch_scans = max(1, int(batch_scans))
        if batch_scans > 1 and not self._groupable(batch_scans):
            batch_scans = 1
        prefetch = max(int(prefetch), readers + 1, 2 * batch_scans if batch_scans > 1 else 0)
        with ThreadPoolExecutor(max_workers=readers, thread_name_prefix="mvlm-ingest") as pool:
            pending = [pool.submit(ingest, f) for f in files[:prefetch]]
            group: list = []  # (file, mesh) of scans waiting to share a network pass

            def flush():
                if not group:
                    return
                with self._lock:  # held for the group's GPU section only, never across a yield
                    with self._timer.stage("total"):
                        results = self.predict_meshes_device([m for _, m in group])
                    for (gf, gm), (landmarks, _) in zip(group, results):
                        self._dump_pre_aligned(gm, gf)  # (ingest pre-aligned the scan)
                        self._rays = None
                        self._after_prediction(gf, landmarks)
                done = [(gf, landmarks) for (gf, _), (landmarks, _) in zip(group, results)]
                pool.submit(_drop, [m for _, m in group])
                group.clear()
                yield from done

            for i, f in enumerate(files):
                if i + prefetch < len(files):
                    pending.append(pool.submit(ingest, files[i + prefetch]))
                try:
                    mesh = pending.pop(0).result()  # re-raises the reader's ValueError / FileNotFoundError
                except BaseException:
                    # the one-by-one loop would have delivered every scan before the malformed one: the scans
                    # waiting in the group get their results first, then the error goes to the caller
                    yield from flush()
                    raise
                if mesh is None:
                    yield from flush()
                    print(f"File {f} does not exist")
                    yield f, None
                    continue
                if batch_scans > 1:
                    group.append((f, mesh))
                    del mesh
                    if len(group) == batch_scans or i + 1 == len(files):
                        yield from flush()
                    continue
                with self._lock:
                    self._rays = None
                    with self._timer.stage("total"):
                        landmarks = self._predict_fused(f, mesh=mesh)
                    self._after_prediction(f, landmarks)
                # returning a scan's 10-25 MB of host arrays to the OS costs milliseconds (page
                # unmapping under the GPU driver's MMU notifier): let the reader thread drop them
                # while this thread goes on to the next scan
                pool.submit(_drop, [mesh])
                del mesh
                yield f, landmarks
            yield from flush()

    def _predict_fused(self, file_name: Path, mesh=None):
        from ..utils.prealign import aligned

        t0 = time.perf_counter()
        if mesh is None:
            file_name = self.renderer_3d._check_file(file_name)
            mesh = self.renderer_3d.load_mesh(file_name, load_texture=self._texture_needed())
        else:
            mesh = aligned(mesh, self.pre_align)
        sharded = self.shard_views and parallel.is_distributed()
        if sharded:
            rank, _ = parallel.rank_world()
            poses = self.renderer_3d.generate_3d_transformations() if rank == 0 else None
            poses = parallel.broadcast_array(poses, (int(self.renderer_3d.n_views), 6), device=self.device)
        else:
            poses = self.renderer_3d.generate_3d_transformations()
        self.timings["load"] = time.perf_counter() - t0
        self._dump_pre_aligned(mesh, file_name)
        landmarks, _ = self.predict_mesh_device(mesh, poses)
        return self._to_original(mesh, landmarks)

    def _dump_pre_aligned(self, mesh, file_name):
        """``pre-align.write_pre_aligned`` (utils3d.py:489-494): the transformed surface as a legacy .vtk file."""
        block = self.pre_align
        if not (block and block.get("write_pre_aligned") and getattr(mesh, "to_original", None) is not None):
            return
        from ..utils.prealign import write_pre_aligned

        folder = Path(self.write_pre_aligned_folder) if self.write_pre_aligned_folder else Path(file_name).parent
        write_pre_aligned(mesh, folder / f"{Path(file_name).stem}_pre_transform_mesh.vtk")

    # ---- the reference's numpy slot protocol (general_pipeline.py:83-108) ----------------
    def _predict_slots(self, file_name: Path):
        tm = self._timer
        with tm.stage("render"):
            if isinstance(self.renderer_3d, HipRenderer3D):
                image_stack, transform_stack, pd = self.renderer_3d.multiview_render(file_name, load_texture=self._texture_needed())
            else:
                image_stack, transform_stack, pd = self.renderer_3d.multiview_render(file_name)
        if self.render_image_stack:
            self.visualize_image_stack(image_stack, file_name)
        with tm.stage("prediction"):
            landmark_stack, valid = self.predictor_2d.predict_landmarks_from_images(image_stack)
        landmark_stack = landmark_stack[:, valid, :]
        transform_stack = transform_stack[valid]
        image_stack = image_stack[valid]
        with tm.stage("lines"):
            lines_s, lines_e = self.estimator_3d.estimate_landmark_lines(image_stack, landmark_stack, transform_stack)
        if self.visualize_rays:
            self._rays = (pd, np.asarray(lines_s), np.asarray(lines_e))
        with tm.stage("consensus"):
            landmarks, error = self.estimator_3d.estimate_landmarks_from_lines(landmark_stack, lines_s, lines_e)
        with tm.stage("project"):
            landmarks = self.estimator_3d.project_landmarks_to_surface(pd, landmarks)
        self._say("Landmarks [Error]: ", f"{error:08.6f}", " mm")
        self.last_error = error
        self._dump_pre_aligned(pd, file_name)
        # a mesh handle that went through the config's pre-align block carries its matrix: results go back to
        # the file's coordinates (the rays kept for visualisation stay in the aligned space, with the mesh handle)
        return self._to_original(pd, landmarks)

    def visualize_image_stack(self, image_stack: np.ndarray, file_name: Path, first_index: int = 0):
        """PNG dump of the rendered views (general_pipeline.py:133-146)."""
        from PIL import Image

        save_folder = self.render_image_folder or Path(file_name).parent
        if not Path(save_folder).exists():
            raise ValueError(f"Folder for --visualize-method flag [{save_folder}] does not exist.")
        for i in range(image_stack.shape[0]):
            single_image = np.uint8(image_stack[i, :, :, 0:3] * 255)
            Image.fromarray(single_image).save(Path(save_folder) / f"{Path(file_name).stem}_{first_index + i:02d}.png")
