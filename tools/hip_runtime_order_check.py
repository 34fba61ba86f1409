import sys, numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
    x = torch.zeros(4, device="cuda:0")
from orb_slam_amd import ORBextractor
from orb_slam_amd.synth import synthetic_frame
ex = ORBextractor(1000, 1.2, 8, 20, 7)
k, d = ex(synthetic_frame(752, 480, 1))
print(order, "extract ok", len(k))
import torch
y = torch.ones(8, device="cuda:0").sum().item()
print(order, "torch ok", y)
k2, d2 = ex(synthetic_frame(752, 480, 1))
assert np.array_equal(d, d2)
print(order, "done")
