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

    for name in ("face40", "coarse", "offscreen"):
        sc = SCENES[name]
        want = raster.multiview_render(sc["verts"], sc["tris"], sc["uvs"], sc["tex"], sc["poses"], subpixel_bits=bits)
        np.testing.assert_array_equal(_hip(sc, bits), want)


def test_subpixel_bits_are_checked():
    from mvlm_amd.utils import HipRenderer3D

    with pytest.raises(ValueError):
        HipRenderer3D(n_views=8, verbose=False, subpixel_bits=3)
