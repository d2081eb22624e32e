"""-m gpu: the HIP rasteriser (through the C ABI) against a real OpenGL's rendering of the reference's GL work
(tests/golden/gl_raster.npz; see tests/test_gl_contract.py and tests/gl_contract.py)."""
import numpy as np
import pytest

from gl_contract import compare, load
from test_gl_contract import MEASURED

pytestmark = pytest.mark.gpu

META, SCENES = load()
GL_BITS = META["gl"]["subpixel_bits"]


def _hip(sc, bits):
    from mvlm_amd.utils import HipRenderer3D, Mesh

    r = HipRenderer3D(n_views=len(sc["poses"]), verbose=False, subpixel_bits=bits)
    out = r.render_device(Mesh(sc["verts"], sc["tris"], sc["uvs"], sc["tex"]), sc["poses"]).cpu().numpy()
    r.check()
    return out


@pytest.mark.parametrize("name", sorted(MEASURED))
def test_hip_rasteriser_against_opengl(name):
    """at the GL's own sub-pixel precision: the same bounds the oracle meets, and the oracle itself bit for bit"""
    from oracle import raster

    sc = SCENES[name]
    got = _hip(sc, GL_BITS)
    r = compare(sc, got)
    print(name, r)
    assert r["unexplained"] == 0, r
    clip, texel, depth1 = MEASURED[name]
    assert r["clip"] <= clip and r["texel"] <= texel and r["depth1"] <= depth1, r
    assert r["clip"] + r["texel"] <= 0.001 * r["pixels"]
    want = raster.multiview_render(sc["verts"], sc["tris"], sc["uvs"], sc["tex"], sc["poses"], subpixel_bits=GL_BITS)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name", sorted(n for n, s in SCENES.items() if s["lattice"]))
def test_default_rasteriser_against_opengl_on_lattice_scenes(name):
    """the product's default (8 sub-pixel bits) compared with OpenGL directly where the bits cannot matter"""
    sc = SCENES[name]
    r = compare(sc, _hip(sc, 8))
    assert r["unexplained"] == 0, r
    clip, texel, depth1 = MEASURED[name]
    assert r["clip"] <= clip and r["texel"] <= texel and r["depth1"] <= depth1, r


@pytest.mark.parametrize("bits", [4, 5, 6, 7, 8])
def test_every_subpixel_setting_equals_the_oracle(bits):
    from oracle import raster

    for name in ("face40", "face224", "coarse", "offscreen"):
        sc = SCENES[name]
        want = raster.multiview_render(sc["verts"], sc["tris"], sc["uvs"], sc["tex"], sc["poses"], subpixel_bits=bits)
        np.testing.assert_array_equal(_hip(sc, bits), want)


def test_subpixel_bits_are_checked():
    from mvlm_amd.utils import HipRenderer3D

    with pytest.raises(ValueError):
        HipRenderer3D(n_views=8, verbose=False, subpixel_bits=3)


def test_landmarks_move_little_between_subpixel_settings():
    """What the one numeric choice OpenGL leaves open does to the RESULT: the planted-peak detector end to end (inlier branch)
    with the rasteriser at 8 and at 4 sub-pixel bits.  Measured (profiles/r06_gl_sensitivity.txt): 4-5 % of the pixels and 3-5 % of
    the argmax pixels differ, the landmarks by a median of 0.00-0.07 and at most 0.3-1.7 model units - far above 1e-3, far below
    the detector's own 2.8 units from the truth: "equal to the reference's VTK path within 1e-3" can only be said against a
    named OpenGL; across OpenGL implementations the reference itself moves by this much."""
    from mvlm_amd import config
    from mvlm_amd.pipeline import pipeline_from_config
    from test_planted_cpu import planted_scene

    mesh, pts, sd, poses = planted_scene(n_views=48)
    pipe = pipeline_from_config(config.default_config("DTU3D", "RGB", n_views=48), weights=sd, verbose=False)
    lm = {}
    for bits in (8, 4):
        pipe.renderer_3d.subpixel_bits = bits
        np.random.seed(1)
        lm[bits], err = pipe.predict_mesh_device(mesh, poses)
        assert err < 10.0                        # inlier branch for every landmark at either setting
    d = np.linalg.norm(lm[8] - lm[4], axis=1)
    assert 0.0 < d.max() < 1.5 and np.median(d) < 0.3, (float(np.median(d)), float(d.max()))
    for bits in (8, 4):
        assert np.median(np.linalg.norm(lm[bits] - pts, axis=1)) < 4.0
