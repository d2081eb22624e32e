"""CPU oracle for the mvlm ``predict_one_file`` hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain CPU restatement
(numpy / torch-CPU / one small C file) of the reference algorithm, each function
citing the reference file:line it follows.  It may be imported ONLY by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` - as the checker / the timed CPU baseline, never as the product
path.  The product (``mvlm_amd``) fails loudly when its HIP library is missing
and never falls back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * cnn / maxima / poses / rays / filter / LSQ / one-shot RANSAC: pinned by
    golden vectors generated from the reference's own importable modules
    (tools/make_golden.py -> tests/golden/*.npz).
  * raster (VTK render) and surface snap (vtkCellLocator): the arithmetic lives
    in the un-vendored third-party ``vtk`` package, absent here -> PARITY
    UNPINNED at pixel level; pinned only by convention round-trips and analytic
    cases.
  * prealign (vtkTransform / vtkCenterOfMass of the legacy Utils3D): same
    absent package -> PARITY UNPINNED at the bit level; the homogeneous-matrix
    form of the reference's call sequence.
"""
