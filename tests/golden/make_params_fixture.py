"""Writes tests/golden/reference_params.json: every constant of the reference's params.py
(a module of plain constants -- it imports numpy only, so it can be loaded here) as data.

    python tests/golden/make_params_fixture.py        # needs /root/reference

The fixture is what pins 2d_object_detection_amd/params.py to the reference's own values."""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("reference_params", "/root/reference/params.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
out = {}
for name in sorted(vars(mod)):
    if name.startswith("_") or name == "np":
        continue
    v = getattr(mod, name)
    out[name] = v.tolist() if isinstance(v, np.ndarray) else v
with open(os.path.join(HERE, "reference_params.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", len(out), "constants")
