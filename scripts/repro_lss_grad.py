"""Reproduce tests/test_lss_plain_gpu.py::test_reference_forward_and_backward_through_the_hip_path with toggles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np, torch
from tests.helpers import seeded_state, t
from tests.test_lss_plain_cpu import CFG, SEED
from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot

def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max())

g = np.load(os.path.join(ROOT, "tests", "golden", "lss_golden.npz"))
cuda = torch.device("cuda:0")
for mode in sys.argv[1:] or ["test", "noeval", "evalonly_fwd", "seed", "notf32", "bench"]:
    if mode == "seed":
        torch.manual_seed(1234)
    if mode == "notf32":
        torch.backends.cudnn.allow_tf32 = False
    if mode == "bench":
        torch.backends.cudnn.benchmark = True
    net = seeded_state(LiftSplatShoot(**CFG), SEED).to(cuda)
    x, rots, trans = (t(g[k], cuda) for k in ("l1_x", "l1_rots", "l1_trans"))
    if mode in ("test", "seed", "evalonly_fwd", "notf32", "bench"):
        net.eval()
        with torch.no_grad():
            bev, depth = net(x, rots, trans)
            if mode != "evalonly_fwd":
                vol, _ = net.get_voxels(x, rots, trans)
    net.train()
    xg = x.clone().requires_grad_()
    bev_t, _ = net(xg, rots, trans)
    (bev_t * t(g["l1_w"], cuda)).sum().backward()
    print(mode, "bev_train", rel(bev_t.detach().cpu(), g["l1_bev_train"]), "x_grad", rel(xg.grad.cpu(), g["l1_x_grad"]),
          "w_grad", rel(net.camencode.depthnet.weight.grad.cpu(), g["l1_depthnet_w_grad"]), flush=True)
