"""RCCL sanity on one GPU: communicator of one rank + all-gather through libbayens, with and without torch loaded."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "--torch-first" in sys.argv:
    import torch, torch.distributed  # noqa
    print("torch imported first", torch.__version__, flush=True)
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
uid = ctx.comm_unique_id()
print("uid ok", len(uid), flush=True)
ctx.comm_init(0, 1, uid)
print("comm init ok", flush=True)
a = ctx.to_device(np.arange(12.0).reshape(3, 4))
b = ctx.empty((3, 4))
ctx.allgather_rows(a, b, 3, 4)
ctx.sync()
print("allgather ok", np.array_equal(b.download(), np.arange(12.0).reshape(3, 4)), flush=True)
ctx.comm_destroy()
