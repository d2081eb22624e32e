"""3-D estimator slot: maxima -> rays -> per-landmark consensus -> surface snap.

Drop-in for the reference's ``Estimator3D`` (src/mvlm/utils/estimator3d.py): same
constructor, attributes (``mode``, ``threshold_quantile``, ``threshold_absolute``
are plain attributes other code sets, dlib_pipeline.py:11-12) and methods, with
the three Python loops replaced by HIP kernels (mvlm_amd/csrc/fusion.hip,
surface.hip).  The one-shot RANSAC index draw stays on the host and uses the
global numpy RNG exactly like the reference (:105), so seeding numpy reproduces
the reference's landmarks.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .mesh_io import Mesh
from .render3d import upload_mesh, view_rotations

__all__ = ["HipEstimator3D"]

_MODES = {"quantile": 0, "absolute": 1}


class HipEstimator3D:
    def __init__(self, mode: str = "quantile", threshold_quantile: float = 0.5, threshold_absolute: float = 0.5,
                 device: int = 0, verbose: bool = True):
        self.mode = mode
        self.threshold_quantile = threshold_quantile
        self.threshold_absolute = threshold_absolute
        self.verbose = verbose
        self.ctx = _lib.get_context(device)
        self._upload_stream = None
        self._draw_bufs: dict = {}
        self._rot_bufs: dict = {}
        self._rot_next = 0

    # ---- helpers ----------------------------------------------------------------------
    def _torch(self):
        import torch

        dev = torch.device("cuda", self.ctx.device)
        self.ctx.bind_current_stream(torch, dev)
        return torch, dev

    def upload_rotations(self, rot: np.ndarray):
        """[N,9] float64 view rotations -> device tensor.  A host-to-device copy from ordinary memory waits for the
        work already enqueued on the stream, so the pipeline does this before it enqueues the step's GPU work."""
        torch, dev = self._torch()
        rot = np.ascontiguousarray(rot, dtype=np.float64)
        if not rot.flags.writeable:  # (view_rotations hands out its memoised tables read-only; torch wants to own writable memory)
            rot = rot.copy()
        return torch.from_numpy(rot).to(dev)

    def upload_rotations_async(self, rot: np.ndarray):
        """The same through pinned staging and a copy ENQUEUED on the current stream: the host does not wait for the work
        already there (the fused pipeline calls this right behind the render it has just enqueued; the rays come after the
        render anyway).  Two staging slots, each guarded by the event of its last copy."""
        torch, dev = self._torch()
        rot = np.ascontiguousarray(rot, dtype=np.float64)
        n = int(rot.shape[0])
        slots = self._rot_bufs.get(n)
        if slots is None:
            slots = self._rot_bufs[n] = [[torch.empty((n, 9), dtype=torch.float64).pin_memory(),
                                          torch.empty((n, 9), dtype=torch.float64, device=dev), torch.cuda.Event(), None] for _ in range(2)]
            for sl in slots:
                sl[3] = sl[0].numpy()
        self._rot_next ^= 1
        pinned, dev_buf, event, view = slots[self._rot_next]
        event.synchronize()  # (long done: a step ends with a wait for its results)
        view[...] = rot
        dev_buf.copy_(pinned, non_blocking=True)
        event.record()
        return dev_buf

    def lines_device(self, landmarks_dev, transform_stack, image_size: int = 256, rot: np.ndarray | None = None,
                     rot_dev=None):
        """maxima f32[NL,N,3] on device + host poses -> (starts, ends) f64[NL,N,3] on device."""
        torch, dev = self._torch()
        nl, n = int(landmarks_dev.shape[0]), int(landmarks_dev.shape[1])
        rot = rot_dev if rot_dev is not None else self.upload_rotations(view_rotations(transform_stack) if rot is None else rot)
        starts = torch.empty((nl, n, 3), dtype=torch.float64, device=dev)
        ends = torch.empty_like(starts)
        self.ctx.check(self.ctx.lib.mvlm_estimate_lines(
            self.ctx.handle, C.c_void_p(landmarks_dev.data_ptr()), C.c_void_p(rot.data_ptr()), n, nl, int(image_size),
            C.c_void_p(starts.data_ptr()), C.c_void_p(ends.data_ptr())))
        return starts, ends

    def draw_ransac_indices(self, counts: np.ndarray) -> np.ndarray:
        """The reference draws once per landmark with >= 3 surviving lines, in landmark order, from the
        global numpy RNG (estimator3d.py:105, :174-179).  counts int[NL] -> draws int32[NL,8]."""
        counts = np.asarray(counts)
        if len(counts) and int(counts.min()) == int(counts.max()) and int(counts[0]) >= 3:
            # the usual case (quantile filter, distinct scores: every landmark keeps the same number of lines): ONE call
            # consumes the global RNG exactly as the per-landmark calls do (masked rejection per element, no state
            # carried between calls; tests/test_host_logic.py checks values and final RNG state) in a tenth of the time -
            # in the sharded pipeline every rank waits for rank 0's draws
            return np.random.randint(0, int(counts[0]), size=(len(counts), 8)).astype(np.int32)
        draws = np.zeros((len(counts), 8), dtype=np.int32)
        for lm, k in enumerate(counts):
            k = int(k)
            if k < 3:
                if self.verbose:
                    print("Not enough points for good estimate of landmark lm_no", lm, k)
                continue
            # np.random.choice(range(k), 8, replace=True) (estimator3d.py:105) consumes the global
            # RNG through randint(0, k, size=8); calling that directly skips building range(k)
            draws[lm] = np.random.randint(0, k, size=8)
        return draws

    def expected_counts(self, n_landmarks: int, n_views: int):
        """Survivor count per landmark that the quantile filter yields for distinct, finite scores:
        ``value > np.quantile(values, q)`` (estimator3d.py:140-147) keeps N - 1 - floor(q (N - 1)) of N values.
        None when the counts cannot be known before the scores exist (absolute mode, unknown mode)."""
        if self.mode != "quantile" or not (0.0 <= float(self.threshold_quantile) <= 1.0) or n_views < 1:
            return None
        k = n_views - 1 - int(np.floor(float(self.threshold_quantile) * (n_views - 1)))
        return np.full(n_landmarks, k, dtype=np.int32)

    def plan_draws(self, n_landmarks: int, n_views: int, draws_fn=None, slot: int = 0):
        """Make the RANSAC draws of the coming consensus BEFORE the step's GPU work is enqueued, where that is
        possible, and put them on the device.  The draws need each landmark's survivor count k (the reference draws
        from range(k), estimator3d.py:105), which only exists once the network has finished; fetching it, drawing
        (~0.3 ms of numpy calls) and uploading the table at that point leaves the GPU idle in the middle of the step.
        In quantile mode k is known beforehand unless scores tie or are NaN (``expected_counts``), so the table is
        drawn for that k now; ``consensus_device(plan=...)`` enqueues the solve with it and its ``verify`` compares
        the real counts afterwards - a mismatch rewinds the global RNG to the state saved here and repeats draws +
        solve, so the stream of random numbers consumed, and every result, is the one the synchronous order gives.
        Returns a dict (``expected`` is None when nothing could be planned: absolute mode).  ``slot``: plans that are
        alive at the same time (several scans sharing one network pass) need staging and device buffers of their own."""
        torch, dev = self._torch()
        plan = {"expected": self.expected_counts(n_landmarks, n_views), "rng_state": np.random.get_state(),
                "draws_fn": draws_fn, "draws_dev": None, "ready": None, "consumed": None}
        if plan["expected"] is not None:
            verbose, self.verbose = self.verbose, False  # "Not enough points" is reported by the pass that counts
            try:
                if draws_fn is not None and getattr(draws_fn, "device_result", False):
                    draws = draws_fn(plan["expected"], keep_on_device=True)
                else:
                    draws = (draws_fn or self.draw_ransac_indices)(plan["expected"])
            finally:
                self.verbose = verbose
            if torch.is_tensor(draws):
                # the table came out of a collective on this rank's GPU (sharded pipeline over RCCL): the solve kernel
                # reads it where it is; the collective is ordered before later work of the current stream
                if tuple(draws.shape) != (n_landmarks, 8) or draws.dtype != torch.int32:
                    raise ValueError(f"RANSAC draws must be int32 [{n_landmarks}, 8], got {tuple(draws.shape)} {draws.dtype}")
                plan["draws_dev"], plan["ready"] = draws.contiguous(), None
                return plan
            draws = np.ascontiguousarray(draws, dtype=np.int32)
            if draws.shape != (n_landmarks, 8):
                raise ValueError(f"RANSAC draws must be [{n_landmarks}, 8], got {draws.shape}")
            # pinned staging buffer + a copy stream of its own: the upload neither waits for the work already
            # enqueued on the compute stream nor blocks the host; the solve waits for the event
            if self._upload_stream is None:
                self._upload_stream = torch.cuda.Stream(device=dev)
            key = (n_landmarks, int(slot))
            bufs = self._draw_bufs.get(key)
            if bufs is None:
                # the device table is written by the UPLOAD stream: it is allocated under that stream, i.e. from that stream's
                # pool of torch's caching allocator - a block the compute stream has just freed (and whose last kernel may
                # still be queued there) can then never be handed out for it (round 4: temporaries of a torch op issued
                # right before the first plan of a new landmark count were, and the late kernel overwrote the draws)
                with torch.cuda.stream(self._upload_stream):
                    table = torch.empty((n_landmarks, 8), dtype=torch.int32, device=dev)
                bufs = self._draw_bufs[key] = (torch.empty((n_landmarks, 8), dtype=torch.int32).pin_memory(), table,
                                               torch.cuda.Event(), torch.cuda.Event())
            pinned, dev_buf, event, consumed = bufs
            event.synchronize()  # the previous upload out of this staging buffer has completed
            pinned.copy_(torch.from_numpy(draws))
            with torch.cuda.stream(self._upload_stream):
                # the previous solve that read this device table (compute stream) comes first: a caller that plans the
                # next scan's draws before synchronising must not overwrite a table a queued solve has yet to read
                self._upload_stream.wait_event(consumed)
                dev_buf.copy_(pinned, non_blocking=True)
                event.record(self._upload_stream)
            plan["draws_dev"], plan["ready"], plan["consumed"] = dev_buf, event, consumed
        return plan

    def consensus_device(self, landmarks_dev, starts, ends, draws_fn=None, deferred: bool = False, plan=None,
                         err_out=None, count_out=None):
        """Filter + one-shot RANSAC + LSQ on device.  ``draws_fn(counts) -> int32[NL,8]`` replaces the local
        RNG draw (the sharded pipeline broadcasts rank 0's).  Returns (landmarks f64[NL,3] tensor,
        per-landmark error f64[NL] tensor, counts int32[NL] numpy).

        ``deferred=True`` returns ``(landmarks, error, verify)`` instead and does not wait for the GPU:
        ``verify(counts=None)`` - to be called once, after the caller has enqueued whatever consumes the landmarks -
        takes (or fetches) the survivor counts, and if the draws were made for other counts (``plan_draws``) repeats
        them and the solve IN PLACE and returns True: the caller must then redo the work it had enqueued on the
        landmarks.  ``plan``: the result of ``plan_draws`` made before the step's GPU work (default: planned here).
        ``err_out`` / ``count_out``: device tensors (f64[NL] / i32[NL]) to write into, e.g. views of one buffer the
        caller fetches with a single copy."""
        torch, dev = self._torch()
        if self.mode not in _MODES:
            raise ValueError(f"Unknown mode for line matching in Estimator: {self.mode}")
        nl, n = int(landmarks_dev.shape[0]), int(landmarks_dev.shape[1])
        mask = torch.empty((nl, n), dtype=torch.uint8, device=dev)
        count = count_out if count_out is not None else torch.empty((nl,), dtype=torch.int32, device=dev)
        self.ctx.check(self.ctx.lib.mvlm_consensus_mask(
            self.ctx.handle, C.c_void_p(landmarks_dev.data_ptr()), n, nl, _MODES[self.mode],
            float(self.threshold_quantile), float(self.threshold_absolute), C.c_void_p(mask.data_ptr()),
            C.c_void_p(count.data_ptr())), ValueError)
        out = torch.empty((nl, 3), dtype=torch.float64, device=dev)
        err = err_out if err_out is not None else torch.empty((nl,), dtype=torch.float64, device=dev)
        if plan is None:
            plan = self.plan_draws(nl, n, draws_fn)
        draws_fn = plan["draws_fn"]

        def solve(draws_dev):
            self.ctx.check(self.ctx.lib.mvlm_consensus_solve(
                self.ctx.handle, C.c_void_p(starts.data_ptr()), C.c_void_p(ends.data_ptr()), C.c_void_p(mask.data_ptr()),
                C.c_void_p(count.data_ptr()), C.c_void_p(draws_dev.data_ptr()), n, nl, C.c_void_p(out.data_ptr()),
                C.c_void_p(err.data_ptr())))

        def draw_and_solve(counts):
            draws = np.ascontiguousarray((draws_fn or self.draw_ransac_indices)(counts), dtype=np.int32)
            if draws.shape != (nl, 8):
                raise ValueError(f"RANSAC draws must be [{nl}, 8], got {draws.shape}")
            keep = torch.from_numpy(draws).to(dev)
            solve(keep)

        state = {"counts": None}
        expected = plan["expected"]
        if expected is not None and len(expected) == nl and plan["draws_dev"] is not None:
            if plan["ready"] is not None:
                torch.cuda.current_stream(dev).wait_event(plan["ready"])
            solve(plan["draws_dev"])
            if plan.get("consumed") is not None:
                plan["consumed"].record(torch.cuda.current_stream(dev))  # plan_draws waits for it before reusing the table

            def verify(counts=None) -> bool:
                counts = state["counts"] = np.asarray(count.cpu().numpy() if counts is None else counts)
                if not np.array_equal(counts, expected):
                    np.random.set_state(plan["rng_state"])
                    draw_and_solve(counts)
                    return True
                if self.verbose and expected[0] < 3:
                    for lm in range(nl):
                        print("Not enough points for good estimate of landmark lm_no", lm, int(expected[0]))
                return False
        else:
            state["counts"] = count.cpu().numpy()
            draw_and_solve(state["counts"])

            def verify(counts=None) -> bool:
                return False
        if deferred:
            return out, err, verify
        verify()
        return out, err, state["counts"]

    def project_device(self, mesh: Mesh, landmarks_dev, out=None):
        torch, dev = self._torch()
        if out is None:
            out = torch.empty_like(landmarks_dev)
        handle = upload_mesh(self.ctx, mesh)
        self.ctx.check(self.ctx.lib.mvlm_project_to_surface(self.ctx.handle, handle, C.c_void_p(landmarks_dev.data_ptr()),
                                                            int(landmarks_dev.shape[0]), C.c_void_p(out.data_ptr())))
        return out

    def clip_rays_device(self, mesh: Mesh, starts_dev, ends_dev):
        """float64 [...,3] segments on the GPU -> (ends clipped to the first surface hit, hit mask uint8 [...])."""
        torch, dev = self._torch()
        s, e = starts_dev.contiguous(), ends_dev.contiguous()
        if s.dtype != torch.float64 or e.dtype != torch.float64 or s.shape != e.shape or s.shape[-1] != 3:
            raise ValueError("clip_rays: starts and ends must be float64 arrays of the same [...,3] shape")
        out = torch.empty_like(e)
        hit = torch.empty(tuple(e.shape[:-1]), dtype=torch.uint8, device=dev)
        n = int(e.numel() // 3)
        if n:
            handle = upload_mesh(self.ctx, mesh)
            self.ctx.bind_current_stream(torch, dev)
            self.ctx.check(self.ctx.lib.mvlm_clip_rays_to_mesh(
                self.ctx.handle, handle, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), n, C.c_void_p(out.data_ptr()),
                C.c_void_p(hit.data_ptr())))
        return out, hit

    def clip_rays_to_mesh(self, pd, line_starts: np.ndarray, line_ends: np.ndarray):
        """``RayVisualizer._clip_rays_to_mesh`` (visualization/ray_visualizer.py:172-192) without VTK: every
        ray segment ends at its first intersection with the mesh (rays that miss keep their end).  Along a
        view ray that point is what the view's depth buffer holds, i.e. the depth-aware unprojection of the
        heatmap maximum.  Returns (new_ends f64 like line_ends, hit bool [...])."""
        if not isinstance(pd, Mesh):
            raise TypeError("clip_rays_to_mesh expects the Mesh handle returned by multiview_render")
        torch, dev = self._torch()
        s = torch.from_numpy(np.ascontiguousarray(line_starts, dtype=np.float64)).to(dev)
        e = torch.from_numpy(np.ascontiguousarray(line_ends, dtype=np.float64)).to(dev)
        out, hit = self.clip_rays_device(pd, s, e)
        return out.cpu().numpy(), hit.cpu().numpy().astype(bool)

    @staticmethod
    def mean_error(err_per_landmark: np.ndarray) -> float:
        """sum_error / n_landmarks with the reference's left-to-right accumulation (:180-183)."""
        # (np.cumsum adds strictly left to right - the loop `s = 0; for e: s = s + e` of the reference, not numpy's pairwise sum)
        return float(np.cumsum(np.asarray(err_per_landmark, dtype=np.float64))[-1]) / len(err_per_landmark)

    # ---- the reference's numpy-in / numpy-out slot methods ----------------------------
    def estimate_landmark_lines(self, image_stack: np.ndarray, landmarks_stack: np.ndarray, transform_stack: np.ndarray):
        torch, dev = self._torch()
        lms = torch.from_numpy(np.ascontiguousarray(landmarks_stack, dtype=np.float32)).to(dev)
        s, e = self.lines_device(lms, np.asarray(transform_stack), int(image_stack.shape[1]))
        return s.cpu().numpy(), e.cpu().numpy()

    def estimate_landmarks_from_lines(self, landmark_stack, lines_s, lines_e):
        torch, dev = self._torch()
        lms = torch.from_numpy(np.ascontiguousarray(landmark_stack, dtype=np.float32)).to(dev)
        s = torch.from_numpy(np.ascontiguousarray(lines_s, dtype=np.float64)).to(dev)
        e = torch.from_numpy(np.ascontiguousarray(lines_e, dtype=np.float64)).to(dev)
        out, err, _ = self.consensus_device(lms, s, e)
        return out.cpu().numpy(), self.mean_error(err.cpu().numpy())

    def project_landmarks_to_surface(self, pd, landmarks):
        if not isinstance(pd, Mesh):
            raise TypeError("project_landmarks_to_surface expects the Mesh handle returned by multiview_render")
        torch, dev = self._torch()
        pts = torch.from_numpy(np.ascontiguousarray(landmarks, dtype=np.float64)).to(dev)
        return self.project_device(pd, pts).cpu().numpy()
