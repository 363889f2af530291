"""Config 5 through the plug-in classes against the resident step (bench.py's FieldWorkload): prints both."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesian_ensembling_amd import _lib
ctx = _lib.Context(0)
fw = bench.FieldWorkload(ctx, 32, 1728, 86, 8, 2, "reduced")
c5 = fw.measure(3, 1)
fw.free()
out = bench.api_path_c5(ctx, 32, 36, 48, 86, 8, 2)
out["resident_step_ms"] = c5["ms_per_step"]
out["resident_fits_per_s"] = c5["value"]
out["ratio"] = round(out["api_path_c5_ms_per_step"] / c5["ms_per_step"], 3)
print(json.dumps(out))
