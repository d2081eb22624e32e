"""A hand-made landmark detector in the reference's own architecture, for the "planted peak" tests
(SURVEY.md 8d): weights for MVLMModel (the 817 reference keys) under which landmark channel k of the
FINAL heatmap (conv11's output, after all 138 live convolutions) peaks at the image of a known surface
point P_k in every view that sees it.  With seeded random weights the heatmaps are un-peaked and every
landmark takes the RANSAC-fail branch; with these the consensus runs its inlier branch end to end.

How: the mesh's texture encodes the surface position, R = u and G = v (planar coordinates of the height
field), so a pixel's colour says which surface point it shows, from any view (the render is unlit).
conv1 + bn1 + ReLU (paulsenpredictor.py:404-406) forms hinge features max(R - t_i, 0), max(G - t_i, 0) on a
knot grid t_i = i/16; every residual block (:251-273) gets zero convolution weights, i.e. it is its identity
branch (or the 1x1 channel embedding where in != out), except the first block of each hourglass's lower
pyramid, which cancels its input so that the hourglasses (:301-361) return their input unchanged;
conv5 / conv9 are centre-tap identities, conv6 / conv7 zero;
conv10 (256 -> NL, linear) combines three neighbouring hinges of R and of G into hat functions centred on the
landmark's knots (u_k, v_k) and conv11 passes them through.  Heatmap k = 2 (hat(R - u_k) + hat(G - v_k)),
maximal where the surface point with texture coordinate (u_k, v_k) is drawn.  The blue
channel carries noise with a tiny weight so that equal colours on neighbouring pixels are not exact ties.
Test helper, CPU only (numpy); used by tests/test_planted_cpu.py and tests/test_gpu_parity.py.
"""
from __future__ import annotations

import numpy as np

N_KNOTS = 16  # hinge knots per colour channel: t_i = i / N_KNOTS


def gradient_textured_mesh(grid: int = 120, tex_size: int = 1024, seed: int = 0):
    """The synthetic face-like height field with the position-encoding texture.  Returns a Mesh."""
    from mvlm_amd.utils.synthetic import face_like_mesh

    m = face_like_mesh(grid, 8, seed)
    rs = np.random.RandomState(seed + 1)
    yy, xx = np.mgrid[0:tex_size, 0:tex_size]
    tex = np.empty((tex_size, tex_size, 3), np.uint8)
    tex[..., 0] = np.rint(255.0 * (xx + 0.5) / tex_size)                    # R = u
    tex[..., 1] = np.rint(255.0 * (tex_size - 0.5 - yy) / tex_size)         # G = v (row 0 = top of the picture = v 1)
    tex[..., 2] = rs.randint(0, 256, (tex_size, tex_size))                  # tie-breaking noise
    m.texture = tex
    return m


def landmark_knots(n_landmarks: int, seed: int = 0) -> np.ndarray:
    """Distinct interior knot pairs (i, j), 3 <= i, j <= N_KNOTS - 3, one per landmark."""
    rs = np.random.RandomState(seed)
    pairs = [(i, j) for i in range(3, N_KNOTS - 2) for j in range(3, N_KNOTS - 2)]
    assert n_landmarks <= len(pairs)
    return np.array([pairs[k] for k in rs.permutation(len(pairs))[:n_landmarks]], np.int64)


def surface_points(mesh, knots: np.ndarray) -> np.ndarray:
    """3-D surface points whose texture coordinate is (i/16, j/16): the planted ground truth."""
    v = mesh.verts.astype(np.float64)
    uv = mesh.uvs.astype(np.float64)
    out = np.empty((len(knots), 3))
    for k, (i, j) in enumerate(knots):
        p = np.array([i / N_KNOTS, j / N_KNOTS])
        for t in mesh.tris:
            a, b, c = uv[t[0]], uv[t[1]], uv[t[2]]
            lo, hi = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
            if (p < lo - 1e-12).any() or (p > hi + 1e-12).any():
                continue
            m = np.array([b - a, c - a]).T
            det = np.linalg.det(m)
            if abs(det) < 1e-18:
                continue
            w1, w2 = np.linalg.solve(m, p - a)
            if w1 >= -1e-9 and w2 >= -1e-9 and w1 + w2 <= 1 + 1e-9:
                out[k] = v[t[0]] + w1 * (v[t[1]] - v[t[0]]) + w2 * (v[t[2]] - v[t[0]])
                break
        else:  # pragma: no cover
            raise AssertionError("knot outside the mesh")
    return out


def planted_state_dict(n_landmarks: int, image_mode: str, knots: np.ndarray, dense_eps: float = 0.0,
                       dense_seed: int = 5) -> dict[str, np.ndarray]:
    """The 817-key state dict of the hand-made detector (see the module docstring).
    ``dense_eps`` > 0 (round 5): every convolution weight additionally carries dense Gaussian noise of ``dense_eps`` times its
    He scale, so all 138 convolutions multiply full weight tensors and contribute to the final heatmaps (at 0.003 about +-20 %
    of a peak's height) while the planted peaks still win - the consensus keeps its inlier branch, and an end-to-end
    comparison exercises every layer's arithmetic on activations of a trained network's order of magnitude instead of the
    +-8000 of seeded random weights.  BatchNorm stays the identity: the hourglass's cancelling block needs relu(bn(x)) = x."""
    from mvlm_amd import arch

    c = arch.IMAGE_CHANNELS[image_mode]
    sel = arch.CHANNEL_SELECT[image_mode]
    assert 0 in sel and 1 in sel and 2 in sel, "the detector reads R, G and B"
    r_in, g_in, b_in = sel.index(0), sel.index(1), sel.index(2)
    shapes = arch.state_dict_shapes(n_landmarks, c)
    sd: dict[str, np.ndarray] = {}
    for key, shp in shapes.items():
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[key] = np.array(1, np.int64)
        elif leaf == "running_var":
            sd[key] = np.full(shp, 1.0 - arch.BN_EPS, np.float32)          # BatchNorm = identity
        elif leaf == "weight" and len(shp) == 1:
            sd[key] = np.ones(shp, np.float32)
        else:
            sd[key] = np.zeros(shp, np.float32)
    nk = N_KNOTS
    w1, b1 = sd["conv1.weight"], sd["conv1.bias"]
    for i in range(nk):
        w1[i, r_in, 1, 1] = 1.0
        b1[i] = -i / nk
        w1[nk + i, g_in, 1, 1] = 1.0
        b1[nk + i] = -i / nk
    w1[2 * nk, b_in, 1, 1] = 1.0                                            # the noise plane, as is
    for p, cin in (("conv2", 64), ("conv4", 128)):                           # 1x1 resample = channel embedding
        w = sd[f"{p}.resample.2.weight"]
        w[np.arange(cin), np.arange(cin), 0, 0] = 1.0
    # hourglass: the first block of the lower pyramid (low1 = rb2(max_pool(x)), paulsenpredictor.py:304-306)
    # cancels its input - conv1 = -identity on the first 128 channels (the features are >= 0 and live in
    # channels < 33, so relu(bn(x)) = x and cat(-x, 0, 0) + x = 0) - hence the lower levels carry zeros and
    # the hourglass returns its input unchanged instead of adding blurred, shifted copies of it
    for hg in ("hg1", "hg2"):
        w = sd[f"{hg}.rb2.conv1.weight"]
        w[np.arange(128), np.arange(128), 1, 1] = -1.0
    for p in ("conv5", "conv9"):                                             # centre-tap identity
        w = sd[f"{p}.weight"]
        w[np.arange(256), np.arange(256), 1, 1] = 1.0
    w10 = sd["conv10.weight"]
    for k, (i, j) in enumerate(knots):
        for base, idx in ((0, int(i)), (nk, int(j))):
            w10[k, base + idx - 1, 1, 1] += 1.0
            w10[k, base + idx, 1, 1] += -2.0
            w10[k, base + idx + 1, 1, 1] += 1.0
        w10[k, 2 * nk, 1, 1] = 2e-4                                          # tie-breaking noise
    w11 = sd["conv11.weight"]
    w11[np.arange(n_landmarks), np.arange(n_landmarks), 1, 1] = 1.0
    if dense_eps > 0.0:
        rs = np.random.RandomState(dense_seed)
        for key in sorted(sd):
            v = sd[key]
            if key.endswith(".weight") and v.ndim == 4:
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                v += (dense_eps * np.sqrt(2.0 / fan_in) * rs.standard_normal(v.shape)).astype(np.float32)
    return sd
