"""Pin the CPU oracle against vectors produced by the reference's own code
(tools/make_golden.py -> tests/golden).  CPU only."""
import contextlib
import io
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, seeded_images
from mvlm_amd import arch, weights
from oracle import cnn as ocnn
from oracle import estimator as oest
from oracle import poses as oposes

MODES = {"RGB": 3, "depth": 1, "RGB+depth": 4, "geometry+depth": 2}


def test_state_dict_keys_match_reference():
    table = json.loads((GOLDEN / "state_dict_keys.json").read_text())
    for tag, ref in table.items():
        nl, mode = tag.split("_", 1)
        mine = arch.state_dict_shapes(int(nl), MODES[mode])
        assert len(ref) == 817
        assert sorted(mine.keys()) == sorted(ref.keys())
        assert {k: list(v) for k, v in mine.items()} == ref


def test_conv_flops_match_survey():
    # live graph GFLOP/view
    assert abs(arch.live_conv_flops_per_view(73, 3) / 1e9 - 146.03) < 0.01
    assert abs(arch.live_conv_flops_per_view(84, 4) / 1e9 - 150.64) < 0.01


def test_blocks_bitwise(golden):
    g = golden("cnn_blocks.npz")
    sd = weights.synthetic_state_dict(73, 3, seed=5)
    for tag, prefix in (("rb_64_128", "conv2"), ("rb_256_256", "hg1.rb1")):
        with torch.no_grad():
            y = ocnn.residual_block(sd, prefix, torch.from_numpy(g[f"{tag}_x"])).numpy()
        np.testing.assert_array_equal(y, g[f"{tag}_y"])
    with torch.no_grad():
        y = ocnn.hourglass(sd, "hg2", torch.from_numpy(g["hg_x"])).numpy()
    np.testing.assert_array_equal(y, g["hg_y"])


@pytest.mark.parametrize("nl,mode", [(73, "RGB"), (84, "RGB+depth"), (73, "geometry+depth"), (84, "depth")])
def test_full_model_maxima(golden, nl, mode):
    g = golden("cnn_full.npz")
    tag = f"{nl}_{mode}"
    seed, img_seed = (int(v) for v in g[f"{tag}_seed"])
    sd = weights.synthetic_state_dict(nl, MODES[mode], seed=seed)
    imgs = seeded_images(img_seed, 2)
    lms, valid, heat = ocnn.predict_landmarks_from_images(sd, imgs, arch.CHANNEL_SELECT[mode], batch_size=2,
                                                          return_heatmaps=True)
    np.testing.assert_array_equal(valid, g[f"{tag}_valid"])
    np.testing.assert_array_equal(heat.numpy()[:, :, 5::16, 3::16], g[f"{tag}_heat_sub"])
    np.testing.assert_array_equal(lms, g[f"{tag}_maxima"])


def test_maxima_simple_and_moment(golden):
    g = golden("maxima.npz")
    hm = g["heatmaps"]
    np.testing.assert_array_equal(ocnn.find_heat_map_maxima(hm, "simple"), g["out_simple"])
    np.testing.assert_array_equal(ocnn.find_heat_map_maxima(hm, "moment"), g["out_moment"])
    fast = ocnn.maxima_fast(torch.from_numpy(hm[None]))
    np.testing.assert_array_equal(fast[:, 0, :], g["out_simple"])


@pytest.mark.parametrize("n", [8, 64, 96, 128])
def test_pose_tables(golden, n):
    g = golden("estimator.npz")
    np.random.seed(0)
    mine = oposes.generate_3d_transformations(n)
    assert mine.dtype == g[f"poses_{n}"].dtype
    np.testing.assert_array_equal(mine, g[f"poses_{n}"])


@pytest.mark.parametrize("tag,n", [("8", 8), ("64", 64)])
def test_rays(golden, tag, n):
    g = golden("estimator.npz")
    s, e = oest.estimate_landmark_lines(256, g[f"lines_{tag}_lms"], g[f"poses_{n}"])
    np.testing.assert_array_equal(s, g[f"lines_{tag}_s"])
    np.testing.assert_array_equal(e, g[f"lines_{tag}_e"])


@pytest.mark.parametrize("tag", ["k0", "k1", "k2", "k6", "parallel"])
def test_lsq(golden, tag):
    g = golden("estimator.npz")
    with np.errstate(all="ignore"):
        p = oest.compute_intersection_between_lines(g[f"lsq_{tag}_pa"], g[f"lsq_{tag}_pb"])
    np.testing.assert_array_equal(p, g[f"lsq_{tag}_p"])


@pytest.mark.parametrize("tag", ["q64", "q8", "qfail", "abs", "absfew", "q128x478"])
def test_consensus(golden, tag):
    g = golden("estimator.npz")
    lms, poses = g[f"fuse_{tag}_lms"], g[f"fuse_{tag}_poses"]
    mode = ["quantile", "absolute"][int(g[f"fuse_{tag}_cfg"][0])]
    q, thr = float(g[f"fuse_{tag}_cfg"][1]), float(g[f"fuse_{tag}_cfg"][2])
    s, e = oest.estimate_landmark_lines(256, lms, poses)
    draws = []
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        out, err = oest.estimate_landmarks_from_lines(lms, s, e, mode, q, thr, draws=draws)
    np.testing.assert_array_equal(np.array([k for _, k, _ in draws]), g[f"fuse_{tag}_draw_k"])
    np.testing.assert_array_equal(np.array([d for _, _, d in draws]).reshape(-1, 8), g[f"fuse_{tag}_draws"])
    np.testing.assert_array_equal(out, g[f"fuse_{tag}_out"])
    assert err == float(g[f"fuse_{tag}_err"])
