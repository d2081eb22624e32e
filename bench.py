#!/usr/bin/env python3
"""Benchmark of the mvlm predict_one_file hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1, either form: `python bench.py --gpus N ...` alone starts its N ranks itself (fresh child processes under
  torch.distributed.run; the parent never touches the GPU and relays rank 0's JSON line), or an external
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (RANK set) runs as a rank.

A "step" is one pass of the hot path over one mesh: render all views of a ~100k-triangle textured
face mesh, run the landmark network on every view, take the heatmap maxima, build the rays, fuse
them per landmark and snap the result to the surface.  The mesh, its texture and the weights are
resident in HBM before the timed region starts.

Default workload = the configuration BASELINE.json's metric is quoted on: configs[2],
BU_3DFE-RGB+depth.json (84 landmarks, 4 input channels), 96 views at 256x256 of one mesh.
With N GPUs the SAME 96 views are sharded 96/N per rank (strong scaling - BASELINE's
"96 views sharded 12/GPU across 8x MI355X") with one all-gather of the per-view maxima (RCCL)
before fusion; `--scaling weak` keeps 96 views per GPU instead.  Other BASELINE configurations:
`--config dtu3d-rgb-64 | bu3dfe-depth-8 | dtu3d-geomdepth-96 | mediapipe-478x128`.

roofline.traffic (HBM bytes per launch of the dominant kernel) is measured by the N = 1 invocation itself with three short
rocprofv3 --pmc child passes after the timed region (each at most 90 s, 150 s together; the timed result is logged first) (live_counters: HBM bytes and MFMA-busy of the dominant kernel); --no-live-traffic / MVLM_BENCH_LIVE_TRAFFIC=0 quote the
committed profiles/rNN_traffic.json instead.

Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0         # same guide, "HBM3E peak BW" (spec; ~6.3 TB/s is achievable)

# BASELINE.json configs[i] -> what the bench builds.  `views` is the mesh's view count the config names.
CONFIGS = {
    "bu3dfe-depth-8": dict(index=0, json="BU_3DFE-depth.json", dataset="BU_3DFE", mode="depth", views=8),
    "dtu3d-rgb-64": dict(index=1, json="DTU3D-RGB.json", dataset="DTU3D", mode="RGB", views=64),
    "bu3dfe-rgbd-96": dict(index=2, json="BU_3DFE-RGB+depth.json", dataset="BU_3DFE", mode="RGB+depth", views=96),
    "dtu3d-geomdepth-96": dict(index=3, json="DTU3D-geometry+depth.json", dataset="DTU3D", mode="geometry+depth", views=96),
    # config 5: the 2-D detector is third-party (out of scope); render + dense fusion with synthetic 2-D landmarks
    "mediapipe-478x128": dict(index=4, json=None, dataset=None, mode=None, views=128, landmarks=478),
}
DEFAULT_CONFIG = "bu3dfe-rgbd-96"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def stage(what: str) -> None:
    """A rank's last stage reached, for the launcher's diagnosis when a multi-rank start stalls (launch_ranks): one small file
    per rank in the directory the launcher names; nothing when the bench runs without that launcher."""
    d = os.environ.get("MVLM_BENCH_STATUS_DIR")
    if not d:
        return
    try:
        with open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}"), "w") as f:
            f.write(json.dumps({"stage": what, "t": round(time.time(), 3), "pid": os.getpid(),
                                "local_rank": os.environ.get("LOCAL_RANK"), "backend": os.environ.get("MVLM_BENCH_BACKEND", "nccl"),
                                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                "visible": {k: os.environ.get(k) for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
                                            if os.environ.get(k) is not None}}))
    except OSError:
        pass


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota
    (MVLM_BENCH_CPU_THREADS lowers it; there is no default cap)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    if os.environ.get("MVLM_BENCH_CPU_THREADS"):
        n = min(n, int(os.environ["MVLM_BENCH_CPU_THREADS"]))
    return max(1, n)


def cpu_baseline(mesh, poses, sd, chan_sel, n_sample: int, n_workload: int, shading: str = "texture"):
    """Time the CPU oracle (a port of the reference path; the reference's own VTK renderer
    cannot run here) on a bounded sample of the same workload, on all of this box's host cores."""
    import torch

    from oracle import pipeline as opipe

    cores = host_cores()
    torch.set_num_threads(cores)
    sub = poses[:n_sample]
    timings = {}
    np.random.seed(1)
    log(f"cpu_baseline: {n_sample} views on {cores} threads ...")
    t0 = time.time()
    opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, sub, sd, chan_sel, batch_size=2, timings=timings,
                       shading=shading)
    dt = time.time() - t0
    return {"value": round(n_sample / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} of the workload's {n_workload} views through the CPU oracle (software rasteriser + "
                      f"torch-CPU fp32 network, batch 2 like the reference, + numpy fusion) on all {cores} host "
                      f"threads, {dt:.1f} s; stages " + ", ".join(f"{k} {v:.2f}s" for k, v in timings.items())}


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
        done = sum(1 for _, lm in pipe.predict_files(files, readers=1) if lm is not None)
        torch.cuda.synchronize()
        folder1 = (time.perf_counter() - t0) / done
        t0 = time.perf_counter()
        done = sum(1 for _, lm in pipe.predict_files(files) if lm is not None)
        torch.cuda.synchronize()
        folder = (time.perf_counter() - t0) / done
    log(f"with ingest: one file {1e3 * single:.1f} ms ({load_ms:.1f} ms of it ingest: OBJ parse, the JPEG decoded on the device by a second thread beside it, upload), "
        f"folder of {n_files}: {1e3 * folder:.1f} ms per scan ({1e3 * folder1:.1f} ms with one reader thread)")
    return {"single_file_views_per_s": round(n_views / single, 2), "folder_views_per_s": round(n_views / folder, 2),
            "folder_one_reader_views_per_s": round(n_views / folder1, 2), "unit": "views/s", "files": n_files,
            "jpeg_decode": "device, ahead of the mesh on a second thread (mvlm_texture_from_jpeg)",
            "note": "OBJ (6.4 MB text) parse, upload and the 2048x2048 JPEG texture's decode (on the GPU, byte for byte "
                    "libjpeg's pixels) included; folder = predict_files with the next scans' ingest on reader threads"}


def synthetic_landmark_predictor(mesh, poses, n_landmarks: int, device):
    """BASELINE configs[4]: ground-truth surface points projected into every view + N(0, 0.5 px)
    noise + 10 % uniform outliers, score U(0,1), in the MediaPipe predictor's output format
    (mediapipepredictor.py:46-48) - resident on the GPU, handed over by a PrecomputedPredictor."""
    import torch

    from mvlm_amd.prediction import PrecomputedPredictor
    from mvlm_amd.utils.render3d import view_rotations

    rs = np.random.RandomState(478)
    pts = mesh.verts[rs.choice(mesh.n_verts, n_landmarks, replace=False)].astype(np.float64)
    rot = view_rotations(poses).reshape(-1, 3, 3)
    n = rot.shape[0]
    lms = np.empty((n_landmarks, n, 3), np.float32)
    for v in range(n):
        q = pts @ rot[v].T
        lms[:, v, 1] = (q[:, 0] + 150) / 300 * 256 + rs.normal(0, 0.5, n_landmarks)
        lms[:, v, 0] = 255 - (q[:, 1] + 150) / 300 * 256 + rs.normal(0, 0.5, n_landmarks)
        lms[:, v, 2] = rs.rand(n_landmarks)
    bad = rs.rand(n_landmarks, n) < 0.1
    lms[bad, 0] = rs.uniform(0, 255, bad.sum())
    lms[bad, 1] = rs.uniform(0, 255, bad.sum())
    lms_dev = torch.from_numpy(lms).to(device)
    state = {"lo": 0}

    def device_fn(images):
        lo = state["lo"]
        return lms_dev[:, lo:lo + int(images.shape[0])].contiguous()

    return PrecomputedPredictor(n_landmarks, device_fn=device_fn), state, pts


def committed_traffic(workload_key: str, kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC pass of THIS workload
    (profiles/rNN_traffic.json, written by tools/summarize_pmc.py: 2 x FETCH_SIZE + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the process; a profile of another
    workload says nothing about this one, so anything but an exact workload match gives null."""
    for f in sorted((REPO / "profiles").glob("r*_traffic.json"), reverse=True):
        try:
            rec = json.loads(f.read_text())
        except Exception:  # noqa: BLE001
            continue
        if rec.get("workload") == workload_key and kernel in rec.get("kernels", {}):
            return round(rec["kernels"][kernel]["hbm_bytes_per_launch"]), f.name
    return None, None


def live_counters(variant_name: str | None, argv: list[str]):
    """Counters of THIS invocation's kernels: three separate rocprofv3 `--pmc` child passes (FETCH_SIZE; WRITE_SIZE;
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE - kernel trace only, no other trace domain, as MI355X_MICROARCH.md prescribes) of a
    one-step run of this same script on the same workload, started as ordinary subprocesses after the timed region
    (`rocprofv3 ... -- python3 bench.py ...`).  Returns a dict or (on any failure: no rocprofv3, no counters) {"error": reason}:
      traffic          HBM bytes per launch of the dominant convolution kernel = 2 x FETCH_SIZE + WRITE_SIZE (KB) per dispatch
                       (gfx950 counts half the bytes of wide reads)
      raster_traffic   the same for the five kernels of one mvlm_render together, per render call
      mfma_busy        SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) of the dominant kernel
      source           how it was measured"""
    import csv
    import re
    import shutil
    import signal
    import subprocess
    import tempfile

    PASS_LIMIT = float(os.environ.get("MVLM_BENCH_PMC_PASS_LIMIT", "90"))
    BUDGET = float(os.environ.get("MVLM_BENCH_PMC_BUDGET", "150"))
    t_budget_end = time.time() + BUDGET
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        return {"error": "rocprofv3 not found"}
    cfg = None  # template arguments of the variant's Cfg<...>, from the variant table itself
    for m in re.finditer(r'X\((\d+), "([^"]+)", Cfg<([^>]*)>\)', (REPO / "mvlm_amd" / "csrc" / "conv_variants.h").read_text()):
        if m[2] == variant_name:
            cfg = m[3]
    if variant_name and cfg is None:
        return {"error": f"no Cfg<> for {variant_name}"}
    RASTER = ("transform_kernel", "classify_kernel", "scan_kernel", "bin_fill_kernel", "tile_kernel")
    dom, ras, n_dom, n_ras = {}, {}, 0, 0   # counter -> sum over the dispatches of the dominant kernel / the rasteriser's kernels
    passes = (("FETCH_SIZE",), ("WRITE_SIZE",)) + ((("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"),) if variant_name else ())
    for counters in passes:
        out = tempfile.mkdtemp(prefix="mvlm_pmc_", dir="/tmp")
        env = dict(os.environ, MVLM_BENCH_NO_INGEST="1", MVLM_BENCH_CHILD="1", TMPDIR="/tmp")
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
               str(Path(__file__).resolve())] + argv + ["--steps", "1", "--warmup", "0", "--cpu-views", "0", "--no-fast-mode"]
        # A pass takes ~3 s; the timed result exists already and must not be lost to a stalled profiler: each pass gets at most
        # PASS_LIMIT seconds and all of them together BUDGET.  rocprofv3 and the python under it run in a session of their
        # own, so that a timeout ends BOTH (killing rocprofv3 alone would leave its python holding the GPU).
        left = min(PASS_LIMIT, t_budget_end - time.time())
        if left <= 1.0:
            shutil.rmtree(out, ignore_errors=True)
            return {"error": f"no time left for the rocprofv3 {'+'.join(counters)} pass ({BUDGET:.0f} s for all passes)"}
        try:
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                child.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(child.pid, sig)
                    except ProcessLookupError:
                        break
                    try:
                        child.communicate(timeout=10)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                shutil.rmtree(out, ignore_errors=True)
                return {"error": f"rocprofv3 {'+'.join(counters)} pass stopped after {left:.0f} s"}
            r = child
        except Exception as e:  # noqa: BLE001
            shutil.rmtree(out, ignore_errors=True)
            return {"error": f"rocprofv3 {'+'.join(counters)} pass failed: {e}"}
        seen_dom = seen_ras = 0
        for f in Path(out).rglob("*counter_collection.csv"):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    c = row.get("Counter_Name")
                    if c not in counters:
                        continue
                    k = row.get("Kernel_Name", "")
                    v = float(row.get("Counter_Value", 0) or 0)
                    if variant_name and f"Cfg<{cfg}>" in k.replace(", false>", ">") and "conv_mfma_kernel" in k and re.search(r">, false(, false)?>", k):
                        dom[c] = dom.get(c, 0.0) + v
                        seen_dom += c == counters[0]
                    elif any(("::" + name + "(") in k or ("::" + name + "<") in k for name in RASTER):
                        ras[c] = ras.get(c, 0.0) + v
                        seen_ras += c == counters[0] and "::tile_kernel" in k
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode != 0 or (variant_name and seen_dom == 0) or seen_ras == 0:
            return {"error": f"rocprofv3 {'+'.join(counters)} pass gave no counters (rc {r.returncode})"}
        n_dom, n_ras = seen_dom, seen_ras  # (every pass runs the same number of launches)
    res = {"raster_traffic": round((2 * ras["FETCH_SIZE"] + ras["WRITE_SIZE"]) * 1024 / n_ras),
           "source": f"live: rocprofv3 --pmc passes of this invocation ({n_dom} dispatches of the kernel, {n_ras} renders per pass)"}
    if variant_name:
        res["traffic"] = round((2 * dom["FETCH_SIZE"] + dom["WRITE_SIZE"]) * 1024 / n_dom)
        gui = dom.get("GRBM_GUI_ACTIVE", 0.0) / 8
        res["mfma_busy"] = round(dom.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 4) if gui > 0 else None
    return res


def visible_gpus() -> int:
    """GPUs a child rank could open, counted from sysfs: no HIP / HSA call, no torch import - this process is the parent of
    the ranks and must not touch the GPU.  A GPU is a KFD topology node with SIMDs whose render node this process may
    open (a container sees the host's whole topology but only its own /dev/dri/renderD*); ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow it as they narrow the runtime's list.  Each rank checks again
    with the runtime itself (authoritative) and exits with the same message when it finds fewer."""
    root = Path("/sys/class/kfd/kfd/topology/nodes")
    if not root.is_dir():
        return 0
    n = 0
    for prop in root.glob("*/properties"):
        try:
            kv = dict(ln.split()[:2] for ln in prop.read_text().splitlines() if len(ln.split()) >= 2)
            if int(kv.get("simd_count", "0")) <= 0:
                continue  # a CPU node
            minor = int(kv.get("drm_render_minor", "-1"))
        except (OSError, ValueError):
            continue  # another container's device: not readable from here
        node = Path(f"/dev/dri/renderD{minor}")
        if minor >= 0 and not (node.exists() and os.access(node, os.R_OK | os.W_OK)):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no RANK in the environment (how the driver starts it): start the N ranks
    as fresh child processes - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same arguments>` - relay rank 0's ONE JSON line to stdout, the
    children's stderr to stderr, and return their exit code.  Nothing here makes a HIP call or imports torch: the GPUs are
    counted from sysfs (visible_gpus)."""
    import queue
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading

    share = os.environ.get("MVLM_BENCH_SHARE_GPU") == "1"
    have = visible_gpus()
    if have < n and not share:
        # sysfs may be masked in a container: before refusing, ask the runtime itself - in a throw-away child, so that this
        # process (the parent of the ranks) still never initialises the GPU
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                               text=True, timeout=600)
            have = max(have, int(r.stdout.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001
            pass
    if have < n and not share:
        log(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have} "
            f"(MVLM_BENCH_SHARE_GPU=1 rehearses the {n}-rank path with every rank on GPU 0 over gloo)")
        return 2
    if have < 1 and not os.environ.get("MVLM_BENCH_TEST_FAULT"):   # (the launcher's own CPU tests start ranks that never reach a GPU call)
        log("bench.py: no GPU visible")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC for the ranks' RCCL (mvlm_amd/parallel.py rccl_environment: the image exports it; a launcher that does not
    # inherit the image's environment must carry it).  Written out here instead of imported: this parent imports nothing
    # that could touch the GPU.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    # The ranks must not outlive the caller's patience silently: the first multi-rank RCCL start on a node is the one step
    # nobody could rehearse, and "killed at the limit, wrote nothing" is its least useful outcome.  The whole run gets a
    # deadline below the driver's ten minutes; each rank's process-group start its own, shorter one (main: init timeout).
    deadline_s = float(os.environ.get("MVLM_BENCH_DEADLINE", "480"))
    status_dir = tempfile.mkdtemp(prefix="mvlm_bench_status_")
    env["MVLM_BENCH_STATUS_DIR"] = status_dir
    log("bench.py: starting", " ".join(cmd), f"(deadline {deadline_s:.0f} s)")
    t_start = time.time()
    # a session of their own: on a deadline the launcher, its ranks and anything they started go together
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    lines: "queue.Queue[str | None]" = queue.Queue()

    def pump():
        for out in proc.stdout:
            lines.put(out.rstrip("\n"))
        lines.put(None)

    threading.Thread(target=pump, daemon=True).start()
    line, timed_out = None, False
    while True:
        left = t_start + deadline_s - time.time()
        if left <= 0:
            timed_out = True
            break
        try:
            out = lines.get(timeout=min(left, 5.0))
        except queue.Empty:
            continue
        if out is None:
            break
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            log(out)  # anything else a rank or the launcher wrote to stdout
    if timed_out:
        log(f"bench.py: the {n} ranks did not finish within {deadline_s:.0f} s (MVLM_BENCH_DEADLINE) - stopping them")
        for sig, wait in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)   # fresh children of this process, in their own session: never this process itself
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=wait)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 124
    else:
        rc = proc.wait()
    if line is not None and not timed_out:
        print(line, flush=True)
    elif rc == 0:
        log("bench.py: the ranks exited cleanly without a result line")
        rc = 1
    if rc != 0:
        diagnose_ranks(status_dir, n, t_start)
    shutil.rmtree(status_dir, ignore_errors=True)
    return rc


def diagnose_ranks(status_dir: str, n: int, t_start: float) -> None:
    """One line per rank on stderr: how far it got (bench.stage), with what backend and device environment - and what to try."""
    for r in range(n):
        try:
            rec = json.loads(Path(status_dir, f"rank{r}").read_text())
            log(f"bench.py: rank {r}: last stage '{rec['stage']}' reached {rec['t'] - t_start:.1f} s after the start; backend {rec['backend']}, "
                f"HSA_ENABLE_IPC_MODE_LEGACY={rec['HSA_ENABLE_IPC_MODE_LEGACY']}, local rank {rec['local_rank']}, device variables {rec['visible'] or 'unset'}, pid {rec['pid']}")
        except (OSError, ValueError, KeyError):
            log(f"bench.py: rank {r}: never reported (the process did not start, or died while importing)")
    log("bench.py: a rank that stops in 'init_process_group (nccl)' or 'first barrier' could not connect RCCL: check that every rank has "
        "HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) and its own GPU; MVLM_BENCH_BACKEND=gloo runs the same sharded step with the two small "
        "collectives staged through the host; MVLM_BENCH_INIT_TIMEOUT / MVLM_BENCH_DEADLINE (seconds) move the two limits")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the config's view count sharded over the ranks (BASELINE), weak = that many per rank")
    ap.add_argument("--views-total", type=int, default=0, help="override the config's view count")
    ap.add_argument("--device-batch", type=int, default=128)
    ap.add_argument("--profile-in-timed", action="store_true",
                    help="record the per-kernel HIP events inside the timed steps (forces the launch-by-launch path, ~2 %% "
                         "slower than the replayed launch graph the product runs); default: time the product path and "
                         "record the events over the same number of steps right after the timed region")
    ap.add_argument("--no-kernel-profile", action="store_true", help=argparse.SUPPRESS)  # old name of the default
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra pass that measures the opt-in fast precision")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="quote roofline.traffic from the committed PMC profile instead of measuring it with three rocprofv3 --pmc child passes")
    ap.add_argument("--precision", default="exact", choices=["exact", "fast", "fast16"], help=argparse.SUPPRESS)  # experiments: time the fast path as the main loop
    ap.add_argument("--selection", default="simple", choices=["simple", "moment"],
                    help="heatmap maxima (paulsenpredictor.py:112-158): the argmax pixel (default) or the 31x31 centroid around it - both fused")
    ap.add_argument("--cnn-execution", default="", help=argparse.SUPPRESS)  # "graphs,concurrency" for experiments, e.g. "1,0"
    ap.add_argument("--cpu-views", type=int, default=-1, help="views in the CPU-baseline sample (-1 = the whole workload, at most 96; 0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started as a plain process: the N ranks are this process's children; it never touches the GPU itself
        sys.exit(launch_ranks(args.gpus))

    # a rank started by an external launcher (the driver's torch.distributed.run) may not have inherited the image's environment:
    # dmabuf IPC for RCCL must be chosen before this process first touches the GPU (mvlm_amd/parallel.py rccl_environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if world > 1 and os.environ.get("MVLM_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < world:
        log(f"bench.py: --gpus {world} needs {world} visible GPUs, the runtime shows {torch.cuda.device_count()}")
        sys.exit(2)
    # MVLM_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a single GPU (gloo, every rank on
    # device 0); the real run is one rank per GPU over RCCL ("nccl" backend on ROCm)
    share_gpu = os.environ.get("MVLM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    use_gloo = share_gpu or os.environ.get("MVLM_BENCH_BACKEND") == "gloo"
    stage("torch imported")
    backend = "none"
    # MVLM_BENCH_FORCE_DIST=1: take the multi-rank path (process group, sharded pipeline, collectives) with whatever
    # world size there is - under torch.distributed.run --nproc-per-node 1 this rehearses RCCL itself on a one-GPU box
    sharded = world > 1 or (os.environ.get("MVLM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if not (sharded and use_gloo):
        torch.cuda.set_device(local_rank)   # (RCCL is created on this rank's device; gloo connects before any GPU call)
        stage("device set")
    if sharded and world == 1:
        os.environ["MVLM_DIST_WORLD_OF_ONE"] = "1"  # mvlm_amd.parallel then runs its collectives in the group of one
    if sharded:
        from datetime import timedelta

        import torch.distributed as dist

        # A process group that cannot connect must say so soon: the default limit of the "nccl" backend is ten minutes - the
        # driver's whole window for this script.  The launcher (launch_ranks) has its own deadline above this one.
        init_timeout = timedelta(seconds=float(os.environ.get("MVLM_BENCH_INIT_TIMEOUT", "120")))
        fault = os.environ.get("MVLM_BENCH_TEST_FAULT", "")   # tests/test_distributed_cpu.py only: "exit:<rank>" / "hang:<rank>"
        if fault.endswith(f":{rank}"):
            stage(f"test fault '{fault}' before init_process_group")
            if fault.startswith("exit"):
                log(f"bench.py: rank {rank}: MVLM_BENCH_TEST_FAULT={fault}: leaving before the process group exists")
                sys.exit(3)
            time.sleep(3600)
        # the communication libraries may log to fd 1 while they connect (gloo does): stdout is reserved for
        # the ONE JSON line, so fd 1 points at stderr until the process group exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            # MVLM_BENCH_BACKEND=gloo: one rank per GPU as always, the two small collectives of a step staged through the host
            # (a way around an RCCL that will not initialise on some node; the default is RCCL = backend "nccl")
            stage(f"init_process_group ({'gloo' if use_gloo else 'nccl'})")
            if use_gloo:
                dist.init_process_group("gloo", timeout=init_timeout)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=init_timeout)
            stage("first barrier")
            dist.barrier()
        except Exception as e:  # noqa: BLE001
            log(f"bench.py: rank {rank}: the process group did not come up within {init_timeout.total_seconds():.0f} s "
                f"(MVLM_BENCH_INIT_TIMEOUT): {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}")
            raise
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        backend = f"{dist.get_backend()} world_size {dist.get_world_size()}"
        stage("process group up")
        if use_gloo:
            torch.cuda.set_device(local_rank)
            stage("device set")

    from mvlm_amd import arch, config, parallel, weights
    from mvlm_amd.utils.synthetic import face_like_mesh

    spec = CONFIGS[args.config]
    views_cfg = args.views_total or spec["views"]
    n_total = views_cfg * world if args.scaling == "weak" else views_cfg
    fusion_only = spec["dataset"] is None
    mesh = face_like_mesh(224, 2048, seed=0)  # 99 458 triangles, 2048^2 texture

    if fusion_only:
        from mvlm_amd import pipeline

        pipe = pipeline.Pipeline(n_views=n_total, device=local_rank, shard_views=sharded, verbose=False)
        nl, c, cfg = spec["landmarks"], 0, None
    else:
        cfg = config.load_config(config.default_config(spec["dataset"], spec["mode"], n_views=n_total))
        pipe = cfg.build_pipeline(weights="synthetic:0", device=local_rank, shard_views=sharded, verbose=False,
                                  device_batch=args.device_batch)
        nl, c = pipe.get_lm_count(), cfg.in_channels
    pipe.renderer_3d.n_views = n_total
    if not fusion_only:
        from mvlm_amd.utils.prealign import aligned, is_active
        from mvlm_amd.utils.synthetic import unaligned_copy

        if is_active(pipe.pre_align):
            # the config file carries a pre-align block (BU_3DFE-depth.json: centre of mass + scale 20): the scan is the
            # raw one that block is written for, brought into the view box the way the product does at load time
            mesh = aligned(unaligned_copy(mesh, pipe.pre_align), pipe.pre_align)
            log(f"pre-align block of {spec['json']} applied: {cfg.pre_align}")
    stage("pipeline built")
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations() if rank == 0 else None
    poses = parallel.broadcast_array(poses, (n_total, 6), device=local_rank)
    lo, hi = parallel.shard_range(n_total, rank, world) if sharded else (0, n_total)
    if fusion_only:
        pred, pred_state, _ = synthetic_landmark_predictor(mesh, poses, nl, torch.device("cuda", local_rank))
        pred_state["lo"] = lo
        pipe.predictor_2d = pred

    def barrier():
        if sharded:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        np.random.seed(1)
        landmarks, err = pipe.predict_mesh_device(mesh, poses)
        return pipe._to_original(mesh, landmarks), err

    if args.cnn_execution and not fusion_only:
        g, cc = (int(v) for v in args.cnn_execution.split(","))
        pipe.predictor_2d.set_execution(graphs=bool(g), concurrency=cc)
    if args.selection != "simple" and not fusion_only:
        pipe.predictor_2d.selection_method = args.selection
    if args.precision != "exact" and not fusion_only:
        pipe.predictor_2d.set_precision(args.precision)
        args.no_fast_mode = True
    cnn_ctx = None if fusion_only else pipe.predictor_2d.ctx
    r_ctx = pipe.renderer_3d.ctx
    # set-up, not warm-up: the product captures its launch graph on the second pass over a set of buffers and replays it
    # from the third; with fewer than three warm-up steps the capture would fall into the timed region
    stage("warm-up steps")
    for _ in range(max(0, 3 - args.warmup)):
        step()
    for _ in range(args.warmup):
        step()
    # Per-kernel HIP events on the launch stream need the launch-by-launch path (two event records around
    # every launch; ~2 % slower than the captured launch graph the product replays).  Default: the timed steps
    # run exactly as the product does and the same number of steps is profiled right after the timed region;
    # --profile-in-timed (N = 1 only) records the events inside the timed steps instead.
    profile_in_timed = world == 1 and args.profile_in_timed
    prof, per_slot = {}, {}
    render_ms, render_calls = [0.0], [0]
    cap = 1024
    slot, var = (C.c_int32 * cap)(), (C.c_int32 * cap)()
    fl, ms = (C.c_double * cap)(), (C.c_float * cap)()
    rv, rverts, rtris, rms = (C.c_int32 * 64)(), (C.c_int32 * 64)(), (C.c_int32 * 64)(), (C.c_float * 64)()

    def set_profiling(on: int):
        if cnn_ctx is not None:
            cnn_ctx.lib.mvlm_cnn_set_profiling(cnn_ctx.handle, on)
        r_ctx.lib.mvlm_render_set_profiling(r_ctx.handle, on)

    def collect():
        # this step's records (event queries only, after the step's final sync)
        if cnn_ctx is not None:
            n = cnn_ctx.lib.mvlm_cnn_get_profile(cnn_ctx.handle, slot, var, fl, ms, cap)
            for i in range(max(n, 0)):
                if slot[i] < 0:
                    continue  # the pool kernel behind a block whose tiles cannot pool (not a convolution)
                p = prof.setdefault(var[i], [0.0, 0.0, 0])
                p[0] += fl[i]
                p[1] += ms[i]
                p[2] += 1
                q = per_slot.setdefault((slot[i], var[i]), [0.0, 0.0, 0])
                q[0] += fl[i]
                q[1] += ms[i]
                q[2] += 1
        n = r_ctx.lib.mvlm_render_get_profile(r_ctx.handle, rv, rverts, rtris, rms, 64)
        for i in range(max(n, 0)):
            render_ms[0] += rms[i]
            render_calls[0] += 1

    if profile_in_timed:
        set_profiling(1)
    # A full collection of the interpreter's oldest generation costs 40 ms in this process (the objects torch's import
    # leaves behind); the set-up above allocates enough to make one due, and at 0.5 ms per step (configs[4]) it then lands
    # in a 20-step timed region and quintuples it.  Collect here, outside; the collector stays ON during the timed steps.
    import gc

    gc.collect()
    stage("barrier before the timed region")
    barrier()
    stage("timed steps")
    step_marks = [] if os.environ.get("MVLM_BENCH_STEP_TIMES") else None  # diagnostic: host time at the end of every timed step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if profile_in_timed:
            collect()
        if step_marks is not None:
            step_marks.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    stage("after the timed region (profiling passes)")
    if step_marks and rank == 0:
        log("timed steps, ms between their returns: " + " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip([t0] + step_marks, step_marks))
            + f"; final wait {1e3 * (t0 + elapsed - step_marks[-1]):.2f}")
    if not profile_in_timed and rank == 0:
        set_profiling(1)
    if sharded:
        parallel.enable_timing(True)  # the all-gather and the draws broadcast of the steps below, per rank
    if not profile_in_timed:
        for _ in range(args.steps):  # every rank joins (collectives inside a step), rank 0 records
            step()
            if rank == 0:
                collect()
    set_profiling(0)
    collectives = parallel.timing_summary() if sharded else None
    if sharded:
        parallel.enable_timing(False)
    # the same shard WITHOUT the sharded path (no collectives, this rank draws for itself), timed on rank 0 while the other
    # ranks wait: what one GPU needs for its 1/N of the views, so that a reader of the JSON line can split the scaling
    # efficiency into "smaller batch per GPU" (single_gpu_shard_ms x N against the N = 1 step) and "cost of sharding"
    # (this rank's ms_per_step against single_gpu_shard_ms)
    single_gpu_shard_ms = None
    if sharded and world > 1:
        barrier()
        if rank == 0 and hi > lo:
            pipe.shard_views = False
            try:
                def shard_step():
                    np.random.seed(1)
                    return pipe.predict_mesh_device(mesh, poses[lo:hi])

                if fusion_only:
                    pred_state["lo"] = lo
                for _ in range(3):
                    shard_step()
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(args.steps):
                    shard_step()
                torch.cuda.synchronize()
                single_gpu_shard_ms = 1e3 * (time.perf_counter() - ts) / args.steps
            finally:
                pipe.shard_views = True
        barrier()
    render_ms, render_calls = render_ms[0], render_calls[0]
    exec_stats = None if fusion_only else pipe.predictor_2d.execution_stats()

    # ---- opt-in "fast" precision (bf16x3 split, mvlm_amd/csrc/conv_fast.hip): a separate figure, never `value` ----------
    fast_mode = fast16_mode = None
    if not sharded and not fusion_only and not args.no_fast_mode:
        p2 = pipe.predictor_2d
        lm_exact, _ = step()
        images = pipe._buffers["images"]
        max_exact = p2.predict_device(images).clone()

        def measure(precision, variant_id, arithmetic):
            p2.set_precision(precision)
            try:
                for _ in range(3):  # launch by launch, capture, first replay
                    step()
                barrier()
                tf0 = time.perf_counter()
                for _ in range(args.steps):
                    lm_fast, _ = step()
                barrier()
                t_fast = (time.perf_counter() - tf0) / args.steps
                max_fast = p2.predict_device(images)
                differ = int((~torch.all(max_fast[:, :, :2] == max_exact[:, :, :2], dim=2)).sum().item())
                fast_prof = {}
                set_profiling(1)
                step()
                n = cnn_ctx.lib.mvlm_cnn_get_profile(cnn_ctx.handle, slot, var, fl, ms, cap)
                fast_slots = {}
                for i in range(max(n, 0)):
                    if slot[i] < 0:
                        continue
                    for q in (fast_prof.setdefault(var[i], [0.0, 0.0, 0]), fast_slots.setdefault((slot[i], var[i]), [0.0, 0.0, 0])):
                        q[0] += fl[i]
                        q[1] += ms[i]
                        q[2] += 1
                r_ctx.lib.mvlm_render_get_profile(r_ctx.handle, rv, rverts, rtris, rms, 64)
                set_profiling(0)
                still = p2.precision
            finally:
                p2.set_precision("exact")
            if rank == 0:
                log(f"{precision} precision, conv kernels of one step:")
                for k, (f, t_ms, cnt) in sorted(fast_prof.items(), key=lambda kv: -kv[1][1]):
                    log(f"  {cnn_ctx.lib.mvlm_conv_variant_name(k).decode():24s} launches/step {cnt:3d}  {t_ms:8.3f} ms/step  "
                        f"{f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s (fp32-equivalent)")
                if os.environ.get("MVLM_BENCH_PER_LAYER"):
                    names = [sl.name for sl in arch.conv_slots(nl, c)]
                    sizes = arch.conv_spatial_sizes()
                    for (sl, v), (f, t_ms, cnt) in sorted(fast_slots.items()):
                        if sl < 0:
                            continue
                        log(f"    slot {sl:3d} {names[sl]:22s} @{sizes[names[sl]]:3d} {cnn_ctx.lib.mvlm_conv_variant_name(v).decode():22s} "
                            f"{t_ms / cnt * 1e3:9.1f} us/launch  {f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s")
            fk = fast_prof.get(variant_id)
            return {"value": round(n_total / t_fast, 2), "unit": "views/s", "ms_per_step": round(1e3 * t_fast, 3),
                    "arithmetic": arithmetic,
                    "argmax_planes_differing_from_exact": differ, "argmax_planes": int(max_exact.shape[0] * max_exact.shape[1]),
                    "max_landmark_deviation_vs_exact_model_units": round(float(np.abs(lm_fast - lm_exact).max()), 6),
                    "fast_kernel_launches_per_step": fk[2] if fk else 0,
                    "fast_kernel_fp32_equivalent_tflops": round(fk[0] / (fk[1] * 1e-3) / 1e12, 1) if fk else None,
                    "fell_back": still != precision}

        fast_mode = measure("fast", 62, "opt-in: 3x3 layers with 16..256 input and >= 64 output channels on 32-pixel rows on bf16x3-split operands "
                                        "(6 of 9 cross products, v_mfma_f32_32x32x16_bf16, fp32 accumulate); everything else exact fp32")
        fast16_mode = measure("fast16", 61, "opt-in: the same layers on f16x2-split operands (3 of 4 cross products, v_mfma_f32_32x32x16_f16, fp32 "
                                            "accumulate, per-layer power-of-two weight scale); everything else exact fp32")
    per_rank_ms = None
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if torch.distributed.get_backend() == "gloo" else "cuda")
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        per_rank_ms = [1e3 * float(v.item()) / args.steps for v in every]
        elapsed = max(float(v.item()) for v in every)  # the job is done when the slowest rank is

    if rank == 0:
        views_per_s = n_total * args.steps / elapsed
        n_local = hi - lo
        workload_key = f"{args.config}:{n_local}v/gpu"
        flops_view = 0.0 if fusion_only else arch.live_conv_flops_per_view(nl, c)
        # rasteriser: HBM-bound.  Algorithmic bytes per view: mesh read V*20 B + T*12 B,
        # framebuffer write 256^2 * 16 B, one 3-byte texel per pixel at most
        ras_bytes_view = mesh.n_verts * 20 + mesh.n_tris * 12 + 256 * 256 * 16 + 256 * 256 * 3
        # the timed result goes to stderr before anything else can delay or lose it (the JSON line follows the profiler passes)
        log(f"timed region: {views_per_s:.2f} views/s, {1e3 * elapsed / args.steps:.3f} ms per step over {args.steps} steps ({n_total} views, {world} rank(s))")
        # HBM traffic of this invocation's kernels by counters (three short rocprofv3 child passes), else the committed profile's
        live = None
        # (not when this process itself runs under a profiler: its preloaded tool library would be inherited by the children)
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if (world == 1 and not args.no_live_traffic and not os.environ.get("MVLM_BENCH_CHILD") and args.precision == "exact"
                and os.environ.get("MVLM_BENCH_LIVE_TRAFFIC", "1") != "0" and not profiled):
            child_args = ["--config", args.config, "--device-batch", str(args.device_batch), "--selection", args.selection]
            if args.views_total:
                child_args += ["--views-total", str(args.views_total)]
            dom_name = cnn_ctx.lib.mvlm_conv_variant_name(max(prof, key=lambda k: prof[k][1])).decode() if prof else None
            t0 = time.time()
            live = live_counters(dom_name, child_args)
            log(f"live counters: {live} ({time.time() - t0:.0f} s)")
            if "error" in live:
                live = {"why_not": live["error"]}
        roof_r = None
        if render_calls:
            avg_ms = render_ms / render_calls
            gbs = ras_bytes_view * n_local / (avg_ms * 1e-3) / 1e9
            tr, rsrc = (live["raster_traffic"], live["source"]) if live and "raster_traffic" in live else committed_traffic(workload_key, "rasteriser")
            roof_r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": tr, "traffic_source": rsrc, "kernel": "rasteriser (5 kernels of one mvlm_render)",
                      "kernel_avg_ms": round(avg_ms, 4), "algorithmic_bytes_per_view": ras_bytes_view,
                      "views_per_launch": n_local}
        roof = None
        if prof:
            lib = cnn_ctx.lib
            dom = max(prof, key=lambda k: prof[k][1])  # dominant conv kernel variant by time
            f, t_ms, cnt = prof[dom]
            achieved = f / (t_ms * 1e-3) / 1e12
            total_f = sum(p[0] for p in prof.values())
            total_ms = sum(p[1] for p in prof.values())
            name = lib.mvlm_conv_variant_name(dom).decode()
            traffic, src = (live["traffic"], live["source"]) if live and "traffic" in live else (None, None)
            if traffic is None:
                traffic, src = committed_traffic(workload_key, name)
                if src and live and live.get("why_not"):
                    src = f"{src} (committed profile; live measurement unavailable: {live['why_not']})"
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
                    "mfma_busy": live.get("mfma_busy") if live else None,
                    "kernel": name, "kernel_avg_ms": round(t_ms / cnt, 4),
                    "kernel_launches_per_step": cnt // args.steps,
                    "kernel_share_of_conv_time": round(t_ms / total_ms, 3),
                    "all_conv_kernels_tflops": round(total_f / (total_ms * 1e-3) / 1e12, 2),
                    "all_conv_frac": round(total_f / (total_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "conv_ms_per_step": round(total_ms / args.steps, 3)}
            for k, (f, t_ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                log(f"  {lib.mvlm_conv_variant_name(k).decode():24s} launches/step {cnt // args.steps:3d}  "
                    f"{t_ms / args.steps:8.3f} ms/step  {f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s")
            if os.environ.get("MVLM_BENCH_PER_LAYER"):
                names = [sl.name for sl in arch.conv_slots(nl, c)]
                sizes = arch.conv_spatial_sizes()
                for (sl, v), (f, t_ms, cnt) in sorted(per_slot.items()):
                    log(f"    slot {sl:3d} {names[sl]:22s} @{sizes[names[sl]]:3d} {lib.mvlm_conv_variant_name(v).decode():22s} "
                        f"{t_ms / cnt * 1e3:9.1f} us/launch  {f / (t_ms * 1e-3) / 1e12:7.2f} TFLOP/s")
        elif roof_r is not None:  # fusion-only workload: the rasteriser is the dominant kernel
            roof, roof_r = roof_r, None
        log("stage seconds (last step):", {k: round(v, 5) for k, v in pipe.timings.items()})
        ingest = None
        if not sharded and not fusion_only and not os.environ.get("MVLM_BENCH_NO_INGEST"):
            ingest = ingest_figures(pipe, n_total)
        cpu = None
        n_cpu = min(n_total, 96) if args.cpu_views < 0 else min(args.cpu_views, n_total)
        if n_cpu > 0 and not fusion_only and world == 1:  # the CPU baseline belongs to the N = 1 line only
            sd = weights.synthetic_state_dict(nl, c, seed=0)
            cpu = cpu_baseline(mesh, poses, sd, arch.CHANNEL_SELECT[cfg.image_channels], n_cpu, n_total,
                               shading="geometry" if "geometry" in cfg.image_channels else "texture")
        per_gpu = f"{n_local}" if world == 1 else f"{n_total // world}" + ("" if n_total % world == 0 else "+")
        what = (f"configs[{spec['index']}] {spec['json']}" if spec["json"] else
                f"configs[{spec['index']}] MediaPipe-shaped fusion stress (synthetic 2-D landmarks, no in-scope CNN)")
        out = {
            "metric": "views/sec (render+CNN+fuse) per mesh, 256x256x96 views" if views_cfg == 96 and not fusion_only
                      else "views/sec (render+CNN+fuse) per mesh, 256x256",
            "value": round(views_per_s, 2),
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("f32" if args.precision == "exact" else ("bf16x3" if args.precision == "fast" else "f16x2") + " (opt-in fast precision: NOT the exact path)") if not fusion_only else "f64",
            "data": "synthetic",
            "config": {"workload": f"{what}: {n_total} views @ 256x256 of one {mesh.n_tris}-triangle textured synthetic "
                                   f"face OBJ, {nl} landmarks" + ("" if fusion_only else f", {c} input channels ({spec['mode']}), seeded random weights"),
                       "name": args.config, "views_total": n_total, "views_per_gpu": per_gpu, "landmarks": nl,
                       "in_channels": c, "triangles": mesh.n_tris, "selection_method": args.selection,
                       "live_conv_gflop_per_view": round(flops_view / 1e9, 2),
                       "parallelism": f"views sharded {per_gpu}/GPU x {world} ({args.scaling} scaling; backend {backend}), "
                                      "1 all-gather of maxima + 1 broadcast of RANSAC draws per mesh" if sharded
                                      else "1 GPU, no collective"},
            "roofline": roof,
            "roofline_rasteriser": roof_r,
            "kernel_events": "HIP events live over the timed steps" if profile_in_timed else
                             "HIP events over the same number of steps run right after the timed region (timed steps = product path, replayed launch graphs)",
            "scaling_breakdown": None if not sharded else {
                "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3),
                                         "all": [round(v, 3) for v in per_rank_ms]},
                "all_gather_ms_per_step": None if collectives["all_gather"]["mean_ms"] is None else round(collectives["all_gather"]["mean_ms"], 4),
                "draws_broadcast_ms_per_step": None if collectives["broadcast"]["mean_ms"] is None else round(collectives["broadcast"]["mean_ms"], 4),
                "collective_clock": collectives["all_gather"]["clock"],
                "single_gpu_shard_ms": None if single_gpu_shard_ms is None else round(single_gpu_shard_ms, 3),
                "shard_views": hi - lo,
                "how_to_read": "rank 0's figures.  single_gpu_shard_ms = this rank's shard run unsharded right after the timed region "
                               "(no process-group traffic): x n_gpus against the N = 1 line's ms_per_step = what the smaller batch per GPU "
                               "costs; ms_per_step against single_gpu_shard_ms = what sharding costs (the two collectives, measured "
                               "over the same number of extra steps, + rank skew = max - min of per_rank_ms_per_step)"},
            "cnn_execution": exec_stats,
            "fast_mode": fast_mode,
            "fast16_mode": fast16_mode,
            "cpu_baseline": cpu,
            "with_ingest": ingest,
        }
        print(json.dumps(out), flush=True)
    if sharded:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
