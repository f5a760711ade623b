"""Image-side pipeline steps of the fusion configs — mirror of
projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py of the reference (``PadMultiViewImage``
:9-60, ``NormalizeMultiviewImage`` :64-98, ``CustomCollect3D`` :202-291,
``RandomScaleImageMultiViewImage`` :295-337), the steps between the decoded camera images and the
``img`` / ``img_metas['lidar2img']`` the detector consumes (bevfusion.py:165-189).

Same class names, constructor arguments, ``results`` keys read and written, and the same 4x4
``lidar2img`` update.  The pixel arithmetic itself is mmcv/OpenCV's in the reference (``mmcv.impad``,
``impad_to_multiple``, ``imnormalize``, ``imresize`` -> ``cv2``; both absent from this image), restated
here in numpy from their published behaviour — PARITY OF THE PIXEL VALUES IS UNPINNED, the
bookkeeping (shapes, keys, matrices) is pinned by tests/golden/make_golden_imgpipe.py:

* normalise: float32 copy, optional BGR->RGB, ``(img - float32(mean)) * float32(1 / float64(std))``;
* resize: bilinear with half-pixel centres, separable, horizontal pass first (``cv2.INTER_LINEAR`` on
  float32); for the configs' scale 0.5 that is the mean of each 2x2 block, each pass rounded once;
* pad: constant ``pad_val`` on the bottom / right.

Photometric distortion (:101-199) is commented out in every NewScenes fusion config and not built.
"""
import numpy as np

from omnihd_amd.mm.registry import PIPELINES

__all__ = ["PadMultiViewImage", "NormalizeMultiviewImage", "RandomScaleImageMultiViewImage", "CustomCollect3D",
           "DataContainer", "imnormalize", "imresize_bilinear", "impad"]


class DataContainer:
    """The two attributes of ``mmcv.parallel.DataContainer`` this path uses."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self._data, self._stack, self._padding_value, self._cpu_only, self._pad_dims = data, stack, padding_value, \
            cpu_only, pad_dims

    data = property(lambda self: self._data)
    cpu_only = property(lambda self: self._cpu_only)
    stack = property(lambda self: self._stack)

    def __repr__(self):
        return f"{self.__class__.__name__}({self._data!r})"


# ---- pixel arithmetic (mmcv / cv2 restated) -------------------------------------------------------------
def imnormalize(img, mean, std, to_rgb=True):
    img = np.array(img, dtype=np.float32, copy=True)
    if to_rgb:
        img = img[..., ::-1]
    inv = (1.0 / np.float64(np.asarray(std).reshape(1, -1))).astype(np.float32)
    return (img - np.asarray(mean, dtype=np.float32).reshape(1, -1)) * inv


def _axis_taps(n_in, n_out):
    """cv2.INTER_LINEAR sample positions: src = (dst + 0.5) * (n_in / n_out) - 0.5, clamped taps."""
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    lo = np.floor(src).astype(np.int64)
    w = (src - lo).astype(np.float32)
    w[lo < 0] = 0.0
    lo0 = np.clip(lo, 0, n_in - 1)
    hi0 = np.clip(lo + 1, 0, n_in - 1)
    return lo0, hi0, w


def imresize_bilinear(img, size):
    """``mmcv.imresize(img, (w, h))`` with the default 'bilinear' interpolation on a float image (H, W, C)."""
    w_out, h_out = size
    x = np.asarray(img, dtype=np.float32)
    lo, hi, w = _axis_taps(x.shape[1], w_out)
    w = w[None, :, None]
    x = x[:, lo] * (np.float32(1) - w) + x[:, hi] * w
    lo, hi, w = _axis_taps(x.shape[0], h_out)
    w = w[:, None, None]
    return x[lo] * (np.float32(1) - w) + x[hi] * w


def impad(img, shape, pad_val=0):
    h, w = shape[:2]
    out = np.full((h, w) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


# ---- pipeline steps -----------------------------------------------------------------------------------------
@PIPELINES.register_module()
class PadMultiViewImage:
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None

    def _target(self, img):
        if self.size is not None:
            return self.size
        d = self.size_divisor
        return int(np.ceil(img.shape[0] / d)) * d, int(np.ceil(img.shape[1] / d)) * d

    def __call__(self, results):
        padded = [impad(img, self._target(img), self.pad_val) for img in results["img"]]
        results["ori_shape"] = [img.shape for img in results["img"]]
        results["img"] = padded
        results["img_shape"] = [img.shape for img in padded]
        results["pad_shape"] = [img.shape for img in padded]
        results["pad_fixed_size"] = self.size
        results["pad_size_divisor"] = self.size_divisor
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(size={self.size}, size_divisor={self.size_divisor}, pad_val={self.pad_val})"


@PIPELINES.register_module()
class NormalizeMultiviewImage:
    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, results):
        results["img"] = [imnormalize(img, self.mean, self.std, self.to_rgb) for img in results["img"]]
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})"


@PIPELINES.register_module()
class RandomScaleImageMultiViewImage:
    """One scale only (the reference asserts it, :304): resize every view and left-multiply ``lidar2img`` by
    diag(s, s, 1, 1)."""

    def __init__(self, scales=[], scale_lidar2img=True):
        self.scales, self.scale_lidar2img = scales, scale_lidar2img
        assert len(self.scales) == 1

    def __call__(self, results):
        s = self.scales[np.random.permutation(range(len(self.scales)))[0]]
        y_size = [int(img.shape[0] * s) for img in results["img"]]
        x_size = [int(img.shape[1] * s) for img in results["img"]]
        scale_factor = np.eye(4)
        scale_factor[0, 0] *= s
        scale_factor[1, 1] *= s
        results["img"] = [imresize_bilinear(img, (x_size[i], y_size[i])) for i, img in enumerate(results["img"])]
        if self.scale_lidar2img:
            results["lidar2img"] = [scale_factor @ m for m in results["lidar2img"]]
        results["img_shape"] = [img.shape for img in results["img"]]
        results["ori_shape"] = [img.shape for img in results["img"]]
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(size={self.scales})"


@PIPELINES.register_module()
class CustomCollect3D:
    """Last step: ``data[key]`` for the requested keys (``None`` when absent) and ``data['img_metas']`` = the meta
    keys present in ``results``, wrapped as a cpu-only container."""

    META_KEYS = ("filename", "ori_shape", "img_shape", "lidar2img", "lidar2cam", "ego2lidar", "depth2img", "cam2img",
                 "pad_shape", "scale_factor", "flip", "pcd_horizontal_flip", "pcd_vertical_flip", "box_mode_3d",
                 "box_type_3d", "img_norm_cfg", "pcd_trans", "sample_idx", "prev_idx", "next_idx", "pcd_scale_factor",
                 "pcd_rotation", "pts_filename", "transformation_3d_flow", "scene_token", "can_bus", "pc_range",
                 "occ_size", "occ_path", "lidar_token", "ego2global_transformation", "lidar2ego_transformation")

    def __init__(self, keys, meta_keys=META_KEYS):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        metas = {k: results[k] for k in self.meta_keys if k in results}
        data = {"img_metas": DataContainer(metas, cpu_only=True)}
        for k in self.keys:
            data[k] = results.get(k)
        return data

    def __repr__(self):
        return f"{self.__class__.__name__}(keys={self.keys}, meta_keys={self.meta_keys})"
