"""bench.py's `paper_script` line on its own (the fourteen collection fits of experiments/full_experiment_script.py:87-113 as one
call): python tools/paper_script.py [n_collections members T]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesian_ensembling_amd import _lib
args = [int(a) for a in sys.argv[1:4]]
print(json.dumps(bench.paper_script_line(_lib.default_context(), *args)), flush=True)
