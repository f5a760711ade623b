"""Shared helpers for the GPU parity tests (inputs are seeded numpy arrays; the oracle is the checker)."""
import ctypes
import os

import numpy as np
import torch

from oracle import lss_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PC_RANGE = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def random_tables(rng, n_vox, n_pix, n_depth, n_points, long_interval=0):
    """Random but well-formed reference-format tables (sorted by ranks_bev, canonical order)."""
    rb = rng.integers(0, n_vox, size=n_points)
    if long_interval:
        rb[:long_interval] = rng.integers(0, n_vox)
    rd = rng.permutation(n_depth)[:n_points] if n_points <= n_depth else rng.integers(0, n_depth, n_points)
    rf = rng.integers(0, n_pix, size=n_points)
    order = np.lexsort((rd, rb))
    rb, rd, rf = rb[order].astype(np.int32), rd[order].astype(np.int32), rf[order].astype(np.int32)
    st, ln = O.run_length(rb)
    return rb, rd, rf, st, ln


def full_size_geometry(tag="r1"):
    """Geometry of the synthetic rig at full size.  rots/trans come from the golden file (they were
    produced by the reference's torch.Tensor(mat).inverse() in the authoring container; LAPACK on
    another host CPU may round the inverse differently, which would move boundary points)."""
    H, W, fx = {"r1": (256, 704, 410.0), "r2": (544, 960, 560.0)}[tag]
    dx, bx, nx = O.gen_dx_bx([PC_RANGE[0], PC_RANGE[3], 0.5], [PC_RANGE[1], PC_RANGE[4], 0.5],
                             [PC_RANGE[2], PC_RANGE[5], 0.5])
    fr = O.create_frustum((H, W), 4, [1, 60, 1])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_golden.npz"))
    geom = O.get_geometry(fr, gold[f"full_{tag}_rots"], gold[f"full_{tag}_trans"])
    return geom, dx, bx, nx


class RefKernels:
    """The reference's own kernels, compiled unmodified by hipcc (oracle/_ref, see oracle/Makefile).
    Host launchers are C++ symbols: bev_pool_v2(int,int,const float*,...) etc."""

    def __init__(self):
        p2 = os.path.join(ROOT, "oracle", "_ref", "libref_bev_pool_v2.so")
        p1 = os.path.join(ROOT, "oracle", "_ref", "libref_bev_pool_v1.so")
        self.ok = os.path.exists(p2) and os.path.exists(p1)
        if self.ok:
            self.v2 = ctypes.CDLL(p2)
            self.v1 = ctypes.CDLL(p1)

    @staticmethod
    def _p(x):
        return ctypes.c_void_p(x.data_ptr())

    def v2_fwd(self, depth, feat, rd, rf, rb, st, ln, out):
        torch.cuda.synchronize()
        self.v2._Z11bev_pool_v2iiPKfS0_PKiS2_S2_S2_S2_Pf(
            ctypes.c_int(feat.size(-1)), ctypes.c_int(st.numel()), self._p(depth), self._p(feat), self._p(rd),
            self._p(rf), self._p(rb), self._p(st), self._p(ln), self._p(out))
        torch.cuda.synchronize()

    def v2_bwd(self, og, depth, feat, rd, rf, rb, st, ln, dg, fg):
        torch.cuda.synchronize()
        self.v2._Z16bev_pool_v2_gradiiPKfS0_S0_PKiS2_S2_S2_S2_PfS3_(
            ctypes.c_int(feat.size(-1)), ctypes.c_int(st.numel()), self._p(og), self._p(depth), self._p(feat),
            self._p(rd), self._p(rf), self._p(rb), self._p(st), self._p(ln), self._p(dg), self._p(fg))
        torch.cuda.synchronize()


def seeded_state(module, seed, by_name=False):
    """Fill every parameter and buffer of ``module`` from one numpy generator, in state-dict order, so
    that a golden script and a test rebuild the same weights without storing them (BatchNorm running
    variances and weights positive; integer buffers and ``frustum`` untouched).  ``by_name``: one generator per
    tensor, seeded by the tensor's NAME, so that the values do not depend on the order of the state dict."""
    import zlib

    import torch
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, v in module.state_dict().items():
            if not v.is_floating_point() or name.endswith("frustum"):
                continue
            if by_name:
                rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
            shape = tuple(v.shape)
            if name.endswith("running_var"):
                a = rng.uniform(0.5, 2.0, shape)
            elif name.endswith("running_mean"):
                a = rng.normal(0.0, 0.2, shape)
            elif v.dim() == 1 and name.endswith("weight"):            # norm scale
                a = rng.uniform(0.5, 1.5, shape)
            elif v.dim() == 1:                                        # biases
                a = rng.normal(0.0, 0.1, shape)
            else:                                                     # conv / linear weights: fan-in scaled
                a = rng.normal(0.0, (2.0 / max(1, int(np.prod(shape[1:])))) ** 0.5, shape)
            v.copy_(torch.from_numpy(a.astype(np.float32)))
    return module
