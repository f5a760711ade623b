"""oracle/pfn_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT.

numpy (float64) restatement of the reference's pillar feature nets:
  * ``PillarFeatureNetV1.forward``   projects/mmdet3d_plugin/rcfusion/voxel_encoders/pillar_encoder.py:378-432
  * ``PFNLayer.forward``             .../rcfusion/voxel_encoders/utils.py:144-181
  * ``RadarPillarFeatureNet.forward`` pillar_encoder.py:91-153 and ``PFNLayer_Radar.forward`` utils.py:229-280
for nets with ONE layer in 'max' mode (what every NewScenes config builds: feat_channels=[64]).
Pinned by the reference's own outputs on fixed inputs (tests/golden/reference_golden.npz, keys g6_*, produced by importing the
reference classes: tests/golden/make_golden.py) in tests/test_oracle.py.  Training-mode BatchNorm (batch statistics over ALL
pillar x slot rows, padded slots included, biased variance — what BatchNorm1d does on the (N, C, M) permuted tensor of
utils.py:161-162) has no reference-held vector: it follows torch's documented definition and is cross-checked against the
torch formulation of the product's module mirror.
"""
import numpy as np

SPATIAL, VELOCITY, SNR = (0, 1, 2, 7, 8, 9, 10, 11), (3, 4, 12, 13), (5, 6, 14, 15)      # utils.py:229-243 (index_select sets)


def decorate(voxels, num_points, coors, voxel_size, pc_range, cluster=True, center=True, distance=False, legacy=True,
             radar=False):
    """(M, P, F) zero-padded pillars -> (M, P, K) decorated and masked point features (pillar_encoder.py:388-427 / :101-149)."""
    f = np.asarray(voxels, dtype=np.float64)
    n = np.asarray(num_points).astype(np.float64).reshape(-1, 1, 1)
    vx, vy = float(np.float32(voxel_size[0])), float(np.float32(voxel_size[1]))
    x_off, y_off = vx / 2 + pc_range[0], vy / 2 + pc_range[1]                              # :368-369
    parts = []
    if cluster:
        parts.append(f[:, :, :3] - f[:, :, :3].sum(1, keepdims=True) / n)                  # :393-397
    base = f
    if center:
        cx = np.asarray(coors)[:, 3].astype(np.float64)[:, None] * vx + x_off               # coors = (b, z, y, x)
        cy = np.asarray(coors)[:, 2].astype(np.float64)[:, None] * vy + y_off
        fc = np.stack([f[:, :, 0] - cx, f[:, :, 1] - cy], -1)
        if legacy:                                                                          # :410-416: f_center is a view of features
            base = np.concatenate([fc, f[:, :, 2:]], -1)
        parts.append(fc)
    if distance:
        parts.append(np.linalg.norm(base[:, :, :3], axis=2, keepdims=True))
    parts.insert(0, base)
    if radar:
        parts.append(base[:, :, 3:7] - base[:, :, 3:7].sum(1, keepdims=True) / n)           # :137-141
    x = np.concatenate(parts, -1)
    mask = (np.arange(x.shape[1])[None, :] < np.asarray(num_points)[:, None])[:, :, None]  # get_paddings_indicator, utils.py:9-29
    return x * mask


def radar_weight(w1, w2, w3, k=16):
    """The three Linear layers of PFNLayer_Radar on their channel subsets as ONE (64, K) matrix."""
    w = np.zeros((w1.shape[0] + w2.shape[0] + w3.shape[0], k))
    r = 0
    for idx, wi in ((SPATIAL, w1), (VELOCITY, w2), (SNR, w3)):
        w[r:r + wi.shape[0], list(idx)] = wi
        r += wi.shape[0]
    return w


def pfn_forward(x, weight, gamma, beta, running_mean=None, running_var=None, eps=1e-3, training=False):
    """Linear (no bias) -> BatchNorm over channels -> ReLU -> max over the slots (utils.py:160-168).
    Returns (out (M, C), mean, biased variance) — the statistics used for the normalisation."""
    y = np.asarray(x, dtype=np.float64) @ np.asarray(weight, dtype=np.float64).T           # (M, P, C)
    if training:
        rows = y.reshape(-1, y.shape[-1])
        mean, var = rows.mean(0), rows.var(0)
    else:
        mean, var = np.asarray(running_mean, dtype=np.float64), np.asarray(running_var, dtype=np.float64)
    z = (y - mean) / np.sqrt(var + eps) * np.asarray(gamma, dtype=np.float64) + np.asarray(beta, dtype=np.float64)
    return np.maximum(z, 0.0).max(1), mean, var
