import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mvlm_amd.utils import HipEstimator3D
from mvlm_amd.utils.synthetic import face_like_mesh
from mvlm_amd.utils.mesh_io import Mesh
m1 = face_like_mesh(224, 64, 0)
m2 = Mesh(verts=m1.verts.copy(), tris=m1.tris.copy())
e3 = HipEstimator3D(verbose=False)
rs = np.random.RandomState(0)
pts = torch.from_numpy(m1.verts[rs.choice(m1.n_verts, 478, replace=False)].astype(np.float64) + 0.3).cuda()
out = torch.empty_like(pts)
for meshes, tag in (((m1, m1), "same mesh (cache hit)"), ((m1, m2), "alternating meshes (cache miss)")):
    for _ in range(10):
        e3.project_device(meshes[0], pts, out=out); e3.project_device(meshes[1], pts, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        e3.project_device(meshes[0], pts, out=out); e3.project_device(meshes[1], pts, out=out)
    b.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        e3.project_device(meshes[0], pts, out=out); e3.project_device(meshes[1], pts, out=out)
    torch.cuda.synchronize()
    print(f"{tag}: {a.elapsed_time(b) * 1e3 / 200:.1f} us per snap (GPU events), {(time.perf_counter() - t0) * 1e6 / 200:.1f} us wall")
