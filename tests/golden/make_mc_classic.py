#!/usr/bin/env python
"""tests/golden/make_mc_classic.py -- the CLASSIC marching-cubes triangle table (Lorensen & Cline 1987 as tabulated by P. Bourke: the table PyMCubes' `marching_cubes`
walks, which the reference calls at lib/core/utils/nerf_utils.py:88) as a fixture: `mc_classic_table.npy`, int8 [256][16], -1 padded.  PyMCubes is not installed in this
image; scikit-image's source tree is (under /opt/conda, not importable by this interpreter), and its `_marching_cubes_lewiner_luts.py` carries the same classic table
(`CASESCLASSIC`, generated from Lewiner's `LookUpTable.h: casesClassic[256][16]`) as base64 text.  This script decodes that text -- data, not code -- and checks that it IS
the classic table in Bourke's corner / edge numbering: every triangle of every case uses only edges whose end corners lie on different sides.
Run here (the GPU box has no /opt/conda): python tests/golden/make_mc_classic.py"""
import base64, os, re
import numpy as np

SRC = "/opt/conda/lib/python3.9/site-packages/skimage/measure/_marching_cubes_lewiner_luts.py"
text = open(SRC).read()
m = re.search(r'CASESCLASSIC = \(256, 16\), """(.*?)"""', text, re.S)
raw = base64.decodebytes(m.group(1).encode("ascii"))
table = np.frombuffer(raw, dtype=np.int8).reshape(256, 16).copy()
EDGES = [[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]]
for case in range(256):
    for e in table[case]:
        if e >= 0:
            a, b = EDGES[e]
            assert ((case >> a) & 1) != ((case >> b) & 1), (case, e)          # Bourke's numbering: an edge of the surface joins corners on different sides
assert (table[0] == -1).all() and (table[255] == -1).all() and table[1, :3].tolist() == [0, 8, 3]
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_classic_table.npy"), table)
print("wrote mc_classic_table.npy;", int((table >= 0).sum()) // 3, "triangles over 256 cases")
