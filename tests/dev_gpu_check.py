"""Developer script (not a test): run one scene through oracle + HIP path on the GPU box and print a report."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as h

def run(cfg, P=None, t=0, bwd=True):
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    t0 = time.time(); o = h.oracle_forward(ins, st); t_or = time.time() - t0
    g = h.gpu_forward_raw(ins, st)
    torch.cuda.synchronize()
    print(f"[{cfg} P={ins['means3D'].shape[0]}] oracle fwd {t_or:.2f}s R={o['num_rendered']} gpuR={g['num_rendered']} V={(o['radii']>0).sum()}")
    try:
        rep = h.compare_forward(o, g); print("  fwd OK", rep)
    except AssertionError as e:
        print("  FWD MISMATCH:", e)
    if bwd:
        from oracle import oracle
        H, W = st["image_height"], st["image_width"]
        grads = h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3, grad_acc_zero=False)
        t0 = time.time(); ob = oracle.backward(o, *grads); t_ob = time.time() - t0
        gb = h.gpu_backward_raw(ins, g, grads)
        torch.cuda.synchronize()
        try:
            rep = h.compare_backward(ob, gb, o); print(f"  bwd OK (oracle {t_ob:.2f}s)", rep)
        except AssertionError as e:
            print("  BWD MISMATCH:", e)
    return ins, st, o, g

if __name__ == "__main__":
    from ex4dgs_amd import build; build.build()
    print(torch.cuda.get_device_name(0))
    run("cfg1")
    run("cfg2", P=20000)
    run("cfg5", P=5000)
    # timing at scale
    for cfg in ("cfg2", "cfg3"):
        ins, st = h.scene_inputs(cfg)
        ins = {k: v.cuda() for k, v in ins.items()}
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            g = h.gpu_forward_raw(ins, st)
            torch.cuda.synchronize(); t1 = time.time()
            H, W = st["image_height"], st["image_width"]
            grads = h.upstream_grads(g["acc"].cpu(), H, W, seed=3)
            grads = [x.cuda() for x in grads]
            torch.cuda.synchronize(); t2 = time.time()
            gb = h.gpu_backward_raw(ins, g, grads)
            torch.cuda.synchronize(); t3 = time.time()
            print(f"[{cfg}] it{it} fwd {1e3*(t1-t0):.2f} ms bwd {1e3*(t3-t2):.2f} ms R={g['num_rendered']}")
