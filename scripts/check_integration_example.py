"""Runs the ctypes example of INTEGRATION.md section 2 verbatim (extracted from the file) on the GPU box."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (same load order as the package: one libamdhip64 in the process)

text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
block = re.search(r"```python\nimport ctypes as C, numpy as np\n(.*?)```", text, re.S).group(0)
code = block[len("```python\n"):-3]
ns = {}
exec(compile(code, "INTEGRATION.md", "exec"), ns)
rs = np.random.RandomState(0)
n = 96
A = rs.normal(size=(n, n)).astype(np.float32)
Qs = ((A + A.T) / 2).astype(np.float32)
betas = np.geomspace(0.05, 5.0, 200)
states, energies = ns["anneal_dense"](Qs, 32, betas, 7)
want = np.einsum("ri,ij,rj->r", states.astype(np.float64), Qs.astype(np.float64), states.astype(np.float64))
assert states.shape == (32, n) and np.allclose(energies, want, rtol=1e-9, atol=1e-6), (energies[:3], want[:3])
print("INTEGRATION.md ctypes example: ok, best E = %.4f" % energies.min())
