#!/usr/bin/env python3
"""Benchmark of the mvlm predict_one_file hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input: render all views of
a ~100k-triangle textured face mesh, run the landmark network on every view, take the heatmap
maxima, build the rays, fuse them per landmark and snap the result to the surface.  The mesh,
its texture and the weights are resident in HBM before the timed region starts.

Workload (BASELINE.json configs[1]): DTU3D-RGB (73 landmarks, 3 input channels), 64 views at
256x256 per GPU.  With N GPUs the mesh gets 64*N views sharded 64 per rank (weak scaling) with
one all-gather of the per-view maxima (RCCL) before fusion.

Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("MVLM_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(mesh, poses, sd, chan_sel, n_sample: int):
    """Time the CPU oracle (a port of the reference path; the reference's own VTK renderer
    cannot run here) on a bounded sample of the same workload, on this box's host cores."""
    import torch

    from oracle import pipeline as opipe

    cores = host_cores()
    torch.set_num_threads(cores)
    sub = poses[:n_sample]
    timings = {}
    np.random.seed(1)
    log(f"cpu_baseline: {n_sample} views on {cores} threads ...")
    t0 = time.time()
    opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, sub, sd, chan_sel, batch_size=2, timings=timings)
    dt = time.time() - t0
    return {"value": round(n_sample / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} of the workload's views through the CPU oracle (software rasteriser + "
                      f"torch-CPU fp32 network, batch 2 like the reference, + numpy fusion), {dt:.1f} s; stages "
                      + ", ".join(f"{k} {v:.2f}s" for k, v in timings.items())}


def ingest_figures(pipe, n_views: int, n_files: int = 12):
    """SURVEY.md 8(d)'s second figure: views/s of predict_one_file *including* OBJ/JPEG ingest and the
    upload (never `value`), and the same for a folder of scans through predict_files, where the
    reader thread hides the ingest of scan i+1 behind the GPU work on scan i."""
    import shutil
    import tempfile

    import torch

    from mvlm_amd.utils.synthetic import write_face_like_obj

    with tempfile.TemporaryDirectory() as td:
        first = write_face_like_obj(Path(td) / "scan0.obj", grid=224, tex_size=2048, seed=0)
        files = [first]
        for i in range(1, n_files):
            f = Path(td) / f"scan{i}.obj"
            shutil.copy(first, f)
            shutil.copy(first.with_suffix(".jpg"), f.with_suffix(".jpg"))
            files.append(f)
        np.random.seed(0)
        pipe.predict_one_file(files[0])  # page cache + first-use costs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        load_ms = 0.0
        for f in files[:3]:
            assert pipe.predict_one_file(f) is not None
            load_ms += 1e3 * pipe.timings.get("load", 0.0) / 3
        torch.cuda.synchronize()
        single = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        done = sum(1 for _, lm in pipe.predict_files(files) if lm is not None)
        torch.cuda.synchronize()
        folder = (time.perf_counter() - t0) / done
    log(f"with ingest: one file {1e3 * single:.1f} ms ({load_ms:.1f} ms of it OBJ+JPEG parse), "
        f"folder of {n_files}: {1e3 * folder:.1f} ms per scan")
    return {"single_file_views_per_s": round(n_views / single, 2), "folder_views_per_s": round(n_views / folder, 2),
            "unit": "views/s", "files": n_files,
            "note": "OBJ (6.4 MB text) + 2048x2048 JPEG parse and upload included; folder = predict_files "
                    "with the next scan's ingest on a reader thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--views-per-gpu", type=int, default=64)
    ap.add_argument("--device-batch", type=int, default=128)
    ap.add_argument("--cpu-views", type=int, default=48, help="views in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--dataset", default="DTU3D")
    ap.add_argument("--image-mode", default="RGB")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    # MVLM_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a single GPU (gloo, every rank on
    # device 0); the real run is one rank per GPU over RCCL ("nccl" backend on ROCm)
    share_gpu = os.environ.get("MVLM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mvlm_amd import arch, config, parallel, weights
    from mvlm_amd.utils.synthetic import face_like_mesh

    cfg = config.load_config(config.default_config(args.dataset, args.image_mode, n_views=args.views_per_gpu * world))
    pipe = cfg.build_pipeline(weights="synthetic:0", device=local_rank, shard_views=world > 1, verbose=False,
                              device_batch=args.device_batch)
    nl, c = pipe.get_lm_count(), cfg.in_channels
    mesh = face_like_mesh(224, 2048, seed=0)  # 99 458 triangles, 2048^2 texture
    n_total = args.views_per_gpu * world
    pipe.renderer_3d.n_views = n_total
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations() if rank == 0 else None
    poses = parallel.broadcast_array(poses)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        np.random.seed(1)
        return pipe.predict_mesh_device(mesh, poses)

    ctx = pipe.predictor_2d.ctx
    for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
        step()
    # per-kernel HIP events on the launch stream, live over the timed region
    ctx.lib.mvlm_cnn_set_profiling(ctx.handle, 1)
    prof = {}
    per_slot = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lm, err = step()
        # collect this step's per-conv records (event queries only, after the step's final sync)
        import ctypes as C
        cap = 1024
        slot = (C.c_int32 * cap)()
        var = (C.c_int32 * cap)()
        fl = (C.c_double * cap)()
        ms = (C.c_float * cap)()
        n = ctx.lib.mvlm_cnn_get_profile(ctx.handle, slot, var, fl, ms, cap)
        for i in range(max(n, 0)):
            p = prof.setdefault(var[i], [0.0, 0.0, 0])
            p[0] += fl[i]
            p[1] += ms[i]
            p[2] += 1
            q = per_slot.setdefault((slot[i], var[i]), [0.0, 0.0, 0])
            q[0] += fl[i]
            q[1] += ms[i]
            q[2] += 1
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.lib.mvlm_cnn_set_profiling(ctx.handle, 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        views_per_s = n_total * args.steps / elapsed
        flops_view = arch.live_conv_flops_per_view(nl, c)
        # dominant conv kernel variant by time
        roof = None
        if prof:
            dom = max(prof, key=lambda k: prof[k][1])
            f, t_ms, cnt = prof[dom]
            achieved = f / (t_ms * 1e-3) / 1e12
            total_f = sum(p[0] for p in prof.values())
            total_ms = sum(p[1] for p in prof.values())
            # HBM bytes per launch of this kernel from the committed rocprofv3 PMC pass of the same
            # command (FETCH_SIZE x2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes); the
            # counters cannot be read from inside this process, so the newest profiles/rNN_traffic.json is used
            traffic = None
            tfiles = sorted((REPO / "profiles").glob("r*_traffic.json"))
            if tfiles:
                rec = json.loads(tfiles[-1].read_text()).get(ctx.lib.mvlm_conv_variant_name(dom).decode())
                if rec:
                    traffic = round(rec["hbm_bytes_per_launch"])
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "kernel": ctx.lib.mvlm_conv_variant_name(dom).decode(),
                    "kernel_avg_ms": round(t_ms / cnt, 4), "kernel_launches_per_step": cnt // args.steps,
                    "kernel_share_of_conv_time": round(t_ms / total_ms, 3),
                    "all_conv_kernels_tflops": round(total_f / (total_ms * 1e-3) / 1e12, 2),
                    "conv_ms_per_step": round(total_ms / args.steps, 3)}
            for k, (f, t_ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                log(f"  {ctx.lib.mvlm_conv_variant_name(k).decode():24s} launches/step {cnt // args.steps:3d}  "
                    f"{t_ms / args.steps:8.3f} ms/step  {f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s")
        if os.environ.get("MVLM_BENCH_PER_LAYER"):
            names = [sl.name for sl in arch.conv_slots(nl, c)]
            sizes = arch.conv_spatial_sizes()
            for (sl, v), (f, t_ms, cnt) in sorted(per_slot.items()):
                log(f"    slot {sl:3d} {names[sl]:22s} @{sizes[names[sl]]:3d} {ctx.lib.mvlm_conv_variant_name(v).decode():22s} "
                    f"{t_ms / cnt * 1e3:9.1f} us/launch  {f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s")
        log("stage seconds (last step):", {k: round(v, 5) for k, v in pipe.timings.items()})
        ingest = None
        if world == 1 and not os.environ.get("MVLM_BENCH_NO_INGEST"):
            ingest = ingest_figures(pipe, n_total)
        cpu = None
        if args.cpu_views > 0:
            sd = weights.synthetic_state_dict(nl, c, seed=0)
            cpu = cpu_baseline(mesh, poses, sd, arch.CHANNEL_SELECT[cfg.image_channels], min(args.cpu_views, n_total))
        out = {
            "metric": "views/sec (render+CNN+fuse) per mesh, 256x256",
            "value": round(views_per_s, 2),
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.dataset}-{args.image_mode}.json, {args.views_per_gpu} views/GPU @ 256x256, "
                                   f"{mesh.n_tris}-triangle textured synthetic face OBJ, {nl} landmarks, seeded "
                                   "random weights",
                       "views_total": n_total, "landmarks": nl, "in_channels": c, "triangles": mesh.n_tris,
                       "live_conv_gflop_per_view": round(flops_view / 1e9, 2),
                       "parallelism": f"views sharded {args.views_per_gpu}/GPU x {world}, 1 all-gather of maxima"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "with_ingest": ingest,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
