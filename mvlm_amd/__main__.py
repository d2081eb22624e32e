"""Command line front end: ``python -m mvlm_amd -p <obj or folder> [-o out] [-n views]``.

Same behaviour as the reference's ``main.py`` (:9-67): collect ``*.obj`` files, run every
available pipeline over them and write ``<stem>_<pipeline>.txt`` (comma-separated [NL,3]
landmarks, ``np.savetxt(..., delimiter=",")``, main.py:62).  Only the pipelines whose 2-D
predictor this build ships are looped ("bu3dfe", "dtu3d"); the viewer flags of the reference
(``--visualize-iter`` / ``--visualize-img``) need VTK and are not available.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np


def shard_files(files, rank: int, world: int):
    """Round-robin share of ``files`` for ``rank`` of ``world`` processes (order preserved)."""
    return [f for i, f in enumerate(files) if i % world == rank]


def main(argv=None) -> int:
    parser = argparse.ArgumentParser(prog="python -m mvlm_amd")
    parser.add_argument("-p", "--path", type=str, required=True)
    parser.add_argument("-o", "--out", type=str, required=False)
    parser.add_argument("-n", "--n-views", type=int, default=None, help="Number of views to render (default: 8, or the config's)")
    parser.add_argument("--visualize-method", action="store_true", help="Dump the rendered views as PNG next to the mesh")
    parser.add_argument("--pipelines", type=str, default="bu3dfe,dtu3d")
    parser.add_argument("-c", "--config", type=str, default=None,
                        help="a Deep-MVLM JSON config (path, or the stem of one of the reference's configs/*.json, e.g. "
                             "BU_3DFE-depth): ONE pipeline built from its inference keys - model, image channels, view count "
                             "(unless -n is given), pose ranges, line filter, pre-align block - instead of --pipelines")
    parser.add_argument("--weights", type=str, default=None,
                        help='checkpoint path, or "synthetic[:seed]" (no checkpoint is reachable offline)')
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--precision", choices=("exact", "fast", "fast16"), default="exact",
                        help="exact: fp32 on the matrix cores (default, the parity path); fast: the big 3x3 layers on "
                             "bf16x3-split operands (fp32-accurate, not bit-identical; DESIGN.md 4.1b)")
    parser.add_argument("--batch-scans", type=int, default=1,
                        help="that many consecutive scans share one pass of the network (higher throughput at few views per "
                             "scan; a near-tied heatmap maximum may resolve differently than in the one-by-one loop)")
    parser.add_argument("--seed", type=int, default=None,
                        help="seed of the global numpy RNG (poses for N != 8 views and the RANSAC draws come from it, as in the "
                             "reference, which never seeds it): makes a run reproducible")
    args = parser.parse_args(argv)
    if args.out is None:
        args.out = args.path
    input_path, path_to_out = Path(args.path), Path(args.out)
    if not input_path.exists():
        print(f"{input_path.as_posix()} does not exist.")
        return 1
    if input_path.is_file():
        if input_path.suffix.lower() != ".obj":
            print(f"{input_path.as_posix()} is not an .obj file.")
            return 1
        obj_files = [input_path]
        if args.out == args.path:
            path_to_out = input_path.parent
    else:
        obj_files = sorted(input_path.glob("*.obj"))
        if len(obj_files) == 0:
            print("Given folder does not contain any .obj files.")
            return 1
    path_to_out.mkdir(parents=True, exist_ok=True)
    # one process per GPU (python -m torch.distributed.run --nproc-per-node N -m mvlm_amd -p folder/): scans are
    # independent, so rank r simply takes every N-th file on its own GPU - no collective, N times the folder rate
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        obj_files = shard_files(obj_files, rank, world)
        if args.device == 0:
            args.device = int(os.environ.get("LOCAL_RANK", "0"))
        print(f"[rank {rank}/{world}] {len(obj_files)} scans on GPU {args.device}")

    from . import pipeline

    extra = {"precision": args.precision} if args.precision != "exact" else {}
    if args.config is not None:
        from . import config as mvlm_config

        src = args.config if Path(args.config).is_file() else mvlm_config.default_config(args.config)
        jobs = [(Path(args.config).stem, lambda: pipeline.pipeline_from_config(
            src, n_views=args.n_views, weights=args.weights, device=args.device,
            **({"render_image_stack": True} if args.visualize_method else {}), **extra))]
    else:
        jobs = [(p, (lambda p=p: pipeline.create_pipeline(p, render_image_stack=args.visualize_method, n_views=args.n_views or 8,
                                                          weights=args.weights, device=args.device, **extra)))
                for p in args.pipelines.split(",") if p]
    for pname, make in jobs:
        print(f"Pipeline: {pname}")
        if args.seed is not None:
            np.random.seed(args.seed)
        dm = make()
        # ingest of the next scans overlaps the GPU work
        for file, landmarks in dm.predict_files(obj_files, batch_scans=args.batch_scans):
            print(f"Current file: {file}")
            if landmarks is None:
                print(f"Landmarks for {file} could not be predicted -> skipping file [{file.stem}] for pipeline {pname}")
                continue
            np.savetxt((path_to_out / f"{file.stem}_{pname}.txt").as_posix(), landmarks, delimiter=",")
    return 0


if __name__ == "__main__":
    sys.exit(main())
