import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers as h
from tests.test_gpu_round4 import _raw_forward, _raw_backward
from ex4dgs_amd import _C, build
build.build(); _C.load()
_C.set_option("geom_debug_arrays", 1)
ins, st = h.scene_inputs("cfg2", P=20000, dir_scale=0.0)
ins = {k: v.cuda() for k, v in ins.items()}
s, sync = _raw_forward(ins, st)
R = sync[0]
cap = int(1.5 * R)
static = {k: v.clone() for k, v in ins.items()}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    _, f = _raw_forward(static, st, settings=s, instance_capacity=cap, assume_no_flow=True)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s3, fwd = _raw_forward(static, st, settings=s, instance_capacity=cap, assume_no_flow=True)
P = 20000
for shift in (0.0, 0.0, 0.05, 0.05, 0.0):
    moved = ins["means3D"] + shift * torch.tensor([1.0, 0.0, 0.0], device="cuda")
    static["means3D"].copy_(moved)
    g.replay(); torch.cuda.synchronize()
    st_words = fwd[0]._status.tolist()
    gv = _C.geom_views(fwd[3], P)
    tiles = gv["tiles_touched"].long()
    radii = fwd[2]
    s4, ref = _raw_forward(dict(ins, means3D=moved), st)
    rv = _C.geom_views(ref[3], P)
    print("shift", shift, "status", st_words[:4], "sum tiles (graph)", int(tiles[radii > 0].sum()), "eager R", ref[0], "radii equal", bool(torch.equal(radii, ref[2])),
          "tiles equal", bool(torch.equal(tiles[radii > 0], rv["tiles_touched"].long()[ref[2] > 0])), "color equal", bool(torch.equal(fwd[1], ref[1])))
