"""Developer profile of the row-segment partition's scatter kernel: cycles per phase and workgroup (ex4d_debug_rows_prof)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd import _C
from tests import helpers as h
from tests.test_gpu_round4 import _raw_forward
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
ins, st = h.scene_inputs(cfg, P=None, t=137)
ins = {k: v.cuda() for k, v in ins.items()}
lib = _C.load()
_C.set_option("rows_probe", 1)
for i in range(6):
    if i == 1:
        buf = (C.c_ulonglong * 8)(); lib.ex4d_debug_rows_prof(buf, 1)
    _raw_forward(ins, st); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)(); lib.ex4d_debug_rows_prof(buf, 1)
n = max(1, buf[7])
names = ["loads+count", "barrier scan", "placement", "width scan", "expansion", "write-out"]
print(cfg, "workgroups per frame", n / 5)
for i, nm in enumerate(names):
    print(f"  {nm:14s} {buf[i] / n:9.0f} cycles per workgroup")
