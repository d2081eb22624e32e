#!/usr/bin/env python3
"""Time the rasteriser alone on the bench mesh: tools/raster_bench.py [views ...] (default 96 128).

Prints, per mesh, view count and shading, the median HIP-event time of one mvlm_render (the library's own events around
its kernels) and a SHA-256 of the rendered stack - two builds of the library that print the same digests render the same
images bit for bit (MVLM_HIP_LIB=<other .so> selects the build).  Under `rocprofv3 --kernel-trace --stats` the trace
gives the split over the kernels."""
import ctypes as C
import hashlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import numpy as np
import torch

from mvlm_amd.utils import HipRenderer3D
from mvlm_amd.utils.synthetic import face_like_mesh


def main():
    views = [int(v) for v in sys.argv[1:]] or [96, 128]
    reps = 30
    # the bench mesh; a coarse one (big triangles); a very coarse one (triangles wider than 64 pixels)
    for grid, tex in ((224, 2048), (40, 256), (6, 64)):
        mesh = face_like_mesh(grid, tex, seed=0)
        for n in views:
            for shading in ("texture", "geometry"):
                r = HipRenderer3D(n_views=n, verbose=False)
                r.shading = shading
                np.random.seed(0)
                poses = r.generate_3d_transformations()
                out = torch.empty((n, 256, 256, 4), dtype=torch.float32, device="cuda")
                for _ in range(3):
                    r.render_device(mesh, poses, out=out)
                r.check()
                lib, h = r.ctx.lib, r.ctx.handle
                lib.mvlm_render_set_profiling(h, 1)
                for _ in range(reps):
                    r.render_device(mesh, poses, out=out)
                r.check()
                nv, nve, nt, ms = (C.c_int32 * 64)(), (C.c_int32 * 64)(), (C.c_int32 * 64)(), (C.c_float * 64)()
                k = lib.mvlm_render_get_profile(h, nv, nve, nt, ms, 64)
                lib.mvlm_render_set_profiling(h, 0)
                t = sorted(ms[i] for i in range(k))
                digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
                print(f"grid {grid:3d} ({mesh.n_tris:6d} triangles) {n:3d} views {shading:8s}: median {1e3 * t[k // 2]:7.1f} us  "
                      f"min {1e3 * t[0]:7.1f} us  per view {1e3 * t[k // 2] / n:6.2f} us   sha256 {digest}", flush=True)


if __name__ == "__main__":
    main()
