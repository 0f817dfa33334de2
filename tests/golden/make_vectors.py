"""Regenerates tests/golden/oracle_vectors.npz from the oracle (run from the repository root):
       python tests/golden/make_vectors.py
The scenarios live in tests/golden_cases.py.  The file holds oracle OUTPUTS only (the reference itself cannot be built in
this environment); inputs are regenerated from their seeds."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import golden_cases  # noqa: E402

out = {}
for name, (orc, _, _) in golden_cases.CASES.items():
    for key, arr in orc().items():
        out["%s.%s" % (name, key)] = np.asarray(arr, np.float32)
np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
print({k: v.shape for k, v in out.items()})
