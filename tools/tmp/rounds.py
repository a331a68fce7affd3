import sys, time, torch
sys.path.insert(0, '.')
import eccoxide_amd as E
from eccoxide_amd import workload as W
eng = E.Engine(0); dev = torch.device('cuda', 0)
for curve, sizes in [("p384r1", [524288, 589824, 393216]), ("bls12_381_g1", [1048576, 1179648, 983040]), ("p256r1", [1048576, 1310720])]:
    for n in sizes:
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=1)).to(dev)
        rs = torch.from_numpy(W.random_scalars(curve, n, seed=2)).to(dev)
        pts, _ = eng.scalarmul_base_t(curve, rs)
        eng.reserve(curve, n, var=True)
        out, fl = eng.scalarmul_var_t(curve, ks, pts)
        torch.cuda.synchronize()
        best = 1e9
        for r in range(4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); eng.scalarmul_var_t(curve, ks, pts, out, fl); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(curve, n, round(best, 3), "ms", round(best / n * 1e6, 3), "ns/unit")
