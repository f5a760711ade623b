"""Radar input format and loader — mirror of the reference's
``LoadRadarPointsMultiSweeps`` (projects/mmdet3d_plugin/datasets/pipelines/loading.py:113-316) and
of the camera-matrix side of ``LoadMultiViewImageFromFiles_newsc`` (:321-400).

On-disk format: one little-endian float32 ``.bin`` per radar sweep, rows of ``load_dim`` = 8 values
``[x, y, z, v_r, power, motion_state, SNR, valid]`` in the SENSOR frame.  Per frame up to
``sweeps_num`` sweeps of each of six radars are merged into rows of ten values
``[x, y, z, vx_comp, vy_comp, power, snr, dt, Vr_comp, radar_id]`` in the LiDAR/ego frame, then
``use_dim`` selects columns and points outside ``pc_range`` are dropped (strict inequalities).

The arithmetic follows the reference line by line (numpy, float32 inputs promoted exactly where the
reference promotes them) so that results are bit-identical; ``merge_radar_sweeps`` is the array-level
core, the class only adds file reading.  The image loader of the same file (``LoadMultiViewImageFromFiles_newsc``)
decodes with PIL and restates OpenCV's undistortion (pixel parity unpinned, bookkeeping pinned); the 4x4 matrix
bookkeeping that feeds ``img_metas['lidar2img']`` is also available on its own in ``half_scale_front_back`` /
``scale_lidar2img``.
"""
import numpy as np
import torch

from omnihd_amd.mm.registry import PIPELINES

RADAR_ID = {"radar_front": 0, "radar_left_front": 1, "radar_right_front": 2, "radar_back": 3, "radar_left_back": 4,
            "radar_right_back": 5}


def quaternion_rotation_matrix(q_wxyz):
    """3x3 rotation of a (w, x, y, z) quaternion, normalised first (pyquaternion ``rotation_matrix``)."""
    q = np.asarray(getattr(q_wxyz, "elements", q_wxyz), dtype=np.float64)
    n2 = float(np.dot(q, q))
    if abs(1.0 - n2) >= 1e-14 and n2 > 0:
        q = q / np.sqrt(n2)
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z]])


def compensate_sweep(points_sweep, sweep, time_diff, radar_id):
    """One sweep (N, 8) float32 in the sensor frame -> (N, 10) float64 rows in the current LiDAR frame.

    Ego-motion compensation of the radial velocity (reference :262-286): the ego velocity is turned
    into the sensor frame, projected on each point's line of sight and added to the measured v_r; the
    compensated radial speed is split into x/y components and rotated into the LiDAR frame."""
    pts = np.copy(points_sweep)
    xyz, vr = pts[:, :3], pts[:, 3]
    r = np.linalg.norm(xyz, axis=1)
    azimuth = np.arctan2(xyz[:, 1], xyz[:, 0])
    elevation = np.arcsin(xyz[:, 2] / r)
    v_ego = np.array(sweep["ego_velocity"]).reshape(-1, 3)
    v_sensor = v_ego @ np.linalg.inv(quaternion_rotation_matrix(sweep["sensor2ego_rotation"])).T
    v_sensor = np.repeat(v_sensor, pts.shape[0], axis=0)
    vr_comp = (v_sensor[:, 0] * np.cos(azimuth) * np.cos(elevation) + v_sensor[:, 1] * np.sin(azimuth) * np.cos(elevation)
               + v_sensor[:, 2] * np.sin(elevation) + vr)
    vx = vr_comp * np.cos(elevation) * np.cos(azimuth)
    vy = vr_comp * np.cos(elevation) * np.sin(azimuth)
    velo = np.concatenate((vx.reshape(-1, 1), vy.reshape(-1, 1), np.zeros((pts.shape[0], 1))), axis=1)
    rot = np.asarray(sweep["sensor2lidar_rotation"])
    velo = (velo @ rot.T)[:, :2]
    pts[:, :3] = pts[:, :3] @ rot.T                       # float64 product stored back as float32, as upstream
    pts[:, :3] += sweep["sensor2lidar_translation"]
    return np.concatenate([pts[:, :3], velo, pts[:, [4, 6]], np.ones((pts.shape[0], 1)) * time_diff,
                           vr_comp.reshape(-1, 1), np.full((pts.shape[0], 1), radar_id)], axis=1)


def merge_radar_sweeps(radars, read_points, sweeps_num=3, load_dim=8):
    """``radars``: {radar name: [sweep dicts, newest first]} -> (M, 10) float64 (reference :229-300)."""
    merged = []
    for key, sweeps in radars.items():
        use = range(len(sweeps)) if len(sweeps) < sweeps_num else range(sweeps_num)
        ts = int(sweeps[0]["timestamp"]) * 1e-6
        for idx in use:
            sweep = sweeps[idx]
            pts = np.copy(read_points(sweep["data_path"])).reshape(-1, load_dim)
            merged.append(compensate_sweep(pts, sweep, ts - int(sweep["timestamp"]) * 1e-6, RADAR_ID[key]))
    return np.concatenate(merged, axis=0)


def merge_radar_sweeps_device(radars, read_points, device, sweeps_num=3, load_dim=8, pc_range=None):
    """The same merge on the GPU (csrc/radar_merge.hip): the raw sweeps are concatenated and uploaded once, the
    per-sweep constants (ego velocity turned into the sensor frame, rotation, translation, time lag, radar id) are
    prepared on the host in float64 exactly as the reference does, one kernel does the per-return arithmetic.
    -> (points (M, 10) fp32 on ``device``, in-range mask or None)."""
    from omnihd_amd import ops
    raws, consts, offsets = [], [], [0]
    for key, sweeps in radars.items():
        use = range(len(sweeps)) if len(sweeps) < sweeps_num else range(sweeps_num)
        ts = int(sweeps[0]["timestamp"]) * 1e-6
        for idx in use:
            sweep = sweeps[idx]
            pts = np.asarray(read_points(sweep["data_path"]), dtype=np.float32).reshape(-1, load_dim)
            v_ego = np.array(sweep["ego_velocity"]).reshape(-1, 3)
            v_sensor = (v_ego @ np.linalg.inv(quaternion_rotation_matrix(sweep["sensor2ego_rotation"])).T)[0]
            consts.append(np.concatenate([v_sensor, np.asarray(sweep["sensor2lidar_rotation"], dtype=np.float64).ravel(),
                                          np.asarray(sweep["sensor2lidar_translation"], dtype=np.float64),
                                          [ts - int(sweep["timestamp"]) * 1e-6, float(RADAR_ID[key])]]))
            raws.append(pts)
            offsets.append(offsets[-1] + len(pts))
    raw = torch.from_numpy(np.concatenate(raws, 0)).to(device)
    return ops.radar_merge(raw, torch.tensor(offsets, dtype=torch.int32, device=device),
                           torch.from_numpy(np.stack(consts)).to(device), pc_range)


class RadarPoints:
    """Minimal point container (reference core/points/radar_points.py:5-28 over mmdet3d BasePoints):
    float32 tensor (N, points_dim), range test with strict inequalities, boolean indexing."""

    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        t = torch.as_tensor(np.asarray(tensor) if not isinstance(tensor, torch.Tensor) else tensor, dtype=torch.float32)
        if t.numel() == 0:
            t = t.reshape((0, points_dim))
        assert t.dim() == 2 and t.size(-1) == points_dim, t.size()
        self.tensor, self.points_dim, self.attribute_dims, self.rotation_axis = t, points_dim, attribute_dims, 2

    def in_range_3d(self, point_range):
        t = self.tensor
        return ((t[:, 0] > point_range[0]) & (t[:, 1] > point_range[1]) & (t[:, 2] > point_range[2])
                & (t[:, 0] < point_range[3]) & (t[:, 1] < point_range[4]) & (t[:, 2] < point_range[5]))

    def __getitem__(self, item):
        return RadarPoints(self.tensor[item], points_dim=self.points_dim, attribute_dims=self.attribute_dims)

    def __len__(self):
        return self.tensor.shape[0]

    def shuffle(self):
        idx = torch.randperm(len(self), device=self.tensor.device)
        self.tensor = self.tensor[idx]
        return idx

    # ---- augmentation hooks: columns 3:5 are the compensated velocity (vx, vy) and move with the geometry
    # (reference core/points/radar_points.py:30-98; none of them is used by the NewScenes fusion pipelines) ----
    def flip(self, bev_direction="horizontal"):
        pos, vel = {"horizontal": (1, 4), "vertical": (0, 3)}[bev_direction]      # mirror y & vy, or x & vx
        self.tensor[:, pos] = -self.tensor[:, pos]
        self.tensor[:, vel] = -self.tensor[:, vel]

    def scale(self, scale_factor):
        self.tensor[:, :3] *= scale_factor
        self.tensor[:, 3:5] *= scale_factor

    def rotate(self, rotation, axis=None):
        """Angle (about ``axis``, default z) or 3x3 matrix; returns the transposed matrix that was right-multiplied."""
        if not isinstance(rotation, torch.Tensor):
            rotation = self.tensor.new_tensor(rotation)
        assert rotation.shape == torch.Size([3, 3]) or rotation.numel() == 1, f"invalid rotation shape {rotation.shape}"
        axis = self.rotation_axis if axis is None else axis
        if rotation.numel() == 1:
            s, c = torch.sin(rotation), torch.cos(rotation)
            rows = {1: [[c, 0, -s], [0, 1, 0], [s, 0, c]], 2: [[c, -s, 0], [s, c, 0], [0, 0, 1]],
                    -1: [[c, -s, 0], [s, c, 0], [0, 0, 1]], 0: [[0, c, -s], [0, s, c], [1, 0, 0]]}
            if axis not in rows:
                raise ValueError("axis should in range")
            rot_mat_T = rotation.new_tensor(rows[axis]).T
        else:
            rot_mat_T = rotation
        self.tensor[:, :3] = self.tensor[:, :3] @ rot_mat_T
        self.tensor[:, 3:5] = self.tensor[:, 3:5] @ rot_mat_T[:2, :2]
        return rot_mat_T

    def in_range_bev(self, point_range):
        t = self.tensor
        return (t[:, 0] > point_range[0]) & (t[:, 1] > point_range[1]) & (t[:, 0] < point_range[2]) & (t[:, 1] < point_range[3])


@PIPELINES.register_module()
class LoadRadarPointsMultiSweeps:
    """Same constructor and ``__call__(results)`` contract as the reference class (:116-316);
    ``file_client_args`` is accepted for config compatibility (only the disk backend exists here)."""

    def __init__(self, load_dim=8, use_dim=(0, 1, 2, 3, 4, 5, 6, 7), sweeps_num=3, file_client_args=None, max_num=300,
                 pc_range=(-72, -56, -3.0, 72, 56, 5.0), test_mode=False, device=None):
        self.load_dim, self.use_dim, self.sweeps_num = load_dim, list(use_dim), sweeps_num
        self.device = device          # e.g. "cuda:0": merge on the GPU, the points never exist on the host
        self.max_num, self.pc_range, self.test_mode = max_num, list(pc_range), test_mode
        if file_client_args and file_client_args.get("backend", "disk") != "disk":
            raise NotImplementedError("only the disk backend is provided")

    @staticmethod
    def _load_points(pts_filename):
        if pts_filename.endswith(".npy"):
            return np.load(pts_filename)
        return np.fromfile(pts_filename, dtype=np.float32)

    def __call__(self, results):
        if self.device is not None:
            pts, mask = merge_radar_sweeps_device(results["radars"], self._load_points, self.device, self.sweeps_num,
                                                  self.load_dim, self.pc_range)
            pts = pts[mask][:, self.use_dim]
            results["points"] = RadarPoints(pts, points_dim=pts.shape[-1], attribute_dims=None)
            return results
        points = merge_radar_sweeps(results["radars"], self._load_points, self.sweeps_num, self.load_dim)
        points = points[:, self.use_dim]
        points = RadarPoints(points, points_dim=points.shape[-1], attribute_dims=None)
        results["points"] = points[points.in_range_3d(self.pc_range)]
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(sweeps_num={self.sweeps_num})"


@PIPELINES.register_module()
class LoadGTDepth:
    """Sparse depth ground truth of the depth-supervised configs (reference :17-63; bevfusion.py:186
    ``dict(type='LoadGTDepth', scale=0.5)``).  Per camera image ``<...>/cameras/<cam>/<file>`` the file
    ``<...>/depth_gt/<cam>/<file>.bin`` holds float32 rows ``[u, v, d]`` in the pixel grid of the STORED image; front
    and back cameras are stored at twice the side cameras' resolution and are halved first (as their images are at load
    time), then everything is scaled by ``scale``.  Coordinates are truncated through int16 exactly as the reference
    does, points outside the ``depth_dim * scale`` map are dropped, later rows overwrite earlier rows of the same pixel
    (numpy assignment order), and at ``scale == 0.5`` the map gets ``pad // 2`` zero rows on top AND bottom
    (540 -> 544 rows: the reference pads the depth map symmetrically although the image is padded at the bottom only —
    kept).  Output ``results['img_depth']``: float32 tensor (n_cams, H, W), what ``generate_guassian_depth_target``
    consumes."""

    def __init__(self, scale, pad=4, scale_factor_frontandback=0.5, depth_dim=(1080, 1920), device=None):
        self.scale_factor_frontandback, self.depth_dim, self.scale, self.pad = scale_factor_frontandback, depth_dim, scale, pad
        self.device = device          # e.g. "cuda:0": upload the sparse rows, build the maps on the GPU (device_prep.py)

    def depth_map(self, cam_depth, cam_dir):
        cam_depth = np.array(cam_depth, dtype=np.float32, copy=True).reshape(-1, 3)
        if cam_dir in ("camera_front", "camera_back"):
            cam_depth[:, :2] = cam_depth[:, :2] * self.scale_factor_frontandback
        cam_depth[:, :2] = cam_depth[:, :2] * self.scale
        uv = cam_depth[:, :2].astype(np.int16)
        dim = (int(self.depth_dim[0] * self.scale), int(self.depth_dim[1] * self.scale))
        out = np.zeros(dim)
        ok = (uv[:, 1] < dim[0]) & (uv[:, 0] < dim[1]) & (uv[:, 1] >= 0) & (uv[:, 0] >= 0)
        out[uv[ok, 1], uv[ok, 0]] = cam_depth[ok, 2]
        if self.scale == 0.5:
            out = np.pad(out, ((self.pad // 2, self.pad // 2), (0, 0)), "constant", constant_values=(0, 0))
        return out

    def __call__(self, results):
        if self.device is not None:
            from .device_prep import device_depth_maps
            names = list(results["filename"])
            rows = [np.fromfile(n.replace("cameras", "depth_gt") + ".bin", dtype=np.float32, count=-1) for n in names]
            results["img_depth"] = device_depth_maps(rows, [n.split("/")[-2] for n in names], self.scale, self.pad,
                                                     self.scale_factor_frontandback, self.depth_dim, self.device)
            return results
        maps = []
        for name in list(results["filename"]):
            rows = np.fromfile(name.replace("cameras", "depth_gt") + ".bin", dtype=np.float32, count=-1)
            maps.append(torch.Tensor(self.depth_map(rows, name.split("/")[-2])))
        results["img_depth"] = torch.stack(maps)
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class LoadOccupancy_Newscenes:
    """Occupancy ground truth: ``results['occ_path']`` is an ``.npz`` whose ``occ_gt`` holds (N, 4) rows
    (x, y, z voxel index, class); scattered into a dense ``occ_size`` grid, 0 = free (reference :66-104)."""

    def __init__(self, use_semantic=True, class_names=None, occ_size=(240, 160, 16)):
        self.use_semantic, self.class_names = use_semantic, class_names
        self.num_classes = len(class_names or ()) + 1
        self.occ_size = list(occ_size)

    @staticmethod
    def gt_to_voxel(gt, num_classes, occ_size):
        voxel = np.zeros(occ_size)
        voxel[gt[:, 0].astype(np.int64), gt[:, 1].astype(np.int64), gt[:, 2].astype(np.int64)] = gt[:, 3]
        return voxel

    def __call__(self, results):
        occ = np.load(results["occ_path"])["occ_gt"].astype(np.float32)
        results["gt_occ"] = self.gt_to_voxel(occ, self.num_classes, self.occ_size)
        return results

    def __repr__(self):
        return self.__class__.__name__


# ---- camera images ------------------------------------------------------------------------------
def undistort_map(K, dist, height, width):
    """Source coordinates (map_x, map_y), float64 (H, W), of ``cv2.undistort(img, K, dist, None, K)``: every output
    pixel is un-projected with K, pushed through the Brown-Conrady model (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]) and
    projected with K again (OpenCV ``initUndistortRectifyMap`` with R = I and the same camera matrix)."""
    d = np.zeros(12, dtype=np.float64)
    dist = np.asarray(dist, dtype=np.float64).reshape(-1)
    if dist.size not in (4, 5, 8, 12):
        raise ValueError(f"distortion vector of {dist.size} coefficients (4, 5, 8 or 12 supported; no tilt terms)")
    d[:dist.size] = dist
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = d
    K = np.asarray(K, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    radial = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r2 * r2
    return fx * xd + cx, fy * yd + cy


def undistort(img, K, dist):
    """``cv2.undistort(img, K, dist, None, K)`` restated: bilinear sampling at ``undistort_map``, zeros outside the image
    (cv2's remap defaults: INTER_LINEAR, BORDER_CONSTANT 0).  cv2 interpolates 8-bit images in 1/32-pixel fixed point;
    this restatement interpolates in float64 and rounds once — PIXEL PARITY WITH OpenCV IS UNPINNED (it is absent from
    the image), the geometry of the map is the published model."""
    h, w = img.shape[:2]
    mx, my = undistort_map(K, dist, h, w)
    x0, y0 = np.floor(mx).astype(np.int64), np.floor(my).astype(np.int64)
    ax, ay = mx - x0, my - y0
    src = np.asarray(img, dtype=np.float64).reshape(h, w, -1)
    out = np.zeros_like(src)
    for dy, wy in ((0, 1 - ay), (1, ay)):
        for dx, wx in ((0, 1 - ax), (1, ax)):
            xs, ys = x0 + dx, y0 + dy
            ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
            tap = src[np.clip(ys, 0, h - 1), np.clip(xs, 0, w - 1)] * ok[..., None]
            out += tap * (wy * wx)[..., None]
    out = out.reshape(img.shape)
    if np.issubdtype(np.asarray(img).dtype, np.integer):
        info = np.iinfo(np.asarray(img).dtype)
        return np.clip(np.rint(out), info.min, info.max).astype(np.asarray(img).dtype)
    return out.astype(np.asarray(img).dtype)


def _read_image_bgr(path):
    """Decoded image in OpenCV's channel order (what ``mmcv.imread(path, 'unchanged')`` returns for colour JPEG/PNG)."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    return a[..., ::-1].copy() if a.ndim == 3 and a.shape[2] >= 3 else a


@PIPELINES.register_module()
class LoadMultiViewImageFromFiles_newsc:
    """Six camera images of a frame (reference :318-405): decode, undistort with the camera's own intrinsics, halve
    the front and back views (stored at twice the side views' resolution) together with their ``lidar2img`` /
    ``cam_intrinsic``, and fill the shape keys the later steps expect.  ``results['img_shape']`` etc. are the shape
    of the (H, W, 3, n_views) stack, as in the reference.  ``read`` lets a caller supply its own decoder."""

    def __init__(self, to_float32=False, color_type="unchanged", read=None):
        self.to_float32, self.color_type = to_float32, color_type
        self.read = read or _read_image_bgr

    def __call__(self, results):
        from .transform_3d import imresize_bilinear
        filename = results["img_filename"]
        views, lidar2img, cam_intrinsic = [], [], []
        half = 0.5
        for i, name in enumerate(filename):
            k = results["cam_intrinsic"][i]
            view = undistort(self.read(name), k[:3, :3], results["cam_distortion"][i])
            if name.split("/")[-2] in ("camera_front", "camera_back"):
                small = imresize_bilinear(view, (int(view.shape[1] * half), int(view.shape[0] * half)))
                if np.issubdtype(view.dtype, np.integer):
                    small = np.clip(np.rint(small), 0, np.iinfo(view.dtype).max).astype(view.dtype)
                view = small
                s = np.eye(4)
                s[0, 0] *= half
                s[1, 1] *= half
                lidar2img.append(s @ results["lidar2img"][i])
                cam_intrinsic.append(s @ k)
            else:
                lidar2img.append(results["lidar2img"][i])
                cam_intrinsic.append(k)
            views.append(view)
        img = np.stack(views, axis=-1)
        results["lidar2img"], results["cam_intrinsic"] = lidar2img, cam_intrinsic
        if self.to_float32:
            img = img.astype(np.float32)
        results["filename"] = filename
        results["img"] = [img[..., i] for i in range(img.shape[-1])]
        results["img_shape"] = results["ori_shape"] = results["pad_shape"] = img.shape
        results["scale_factor"] = 1.0
        channels = 1 if img.ndim < 3 else img.shape[2]
        results["img_norm_cfg"] = dict(mean=np.zeros(channels, dtype=np.float32), std=np.ones(channels, dtype=np.float32),
                                       to_rgb=False)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(to_float32={self.to_float32}, color_type='{self.color_type}')"


# ---- camera matrices that end up in img_metas['lidar2img'] ------------------------------------
def half_scale_front_back(img_filenames, lidar2img, cam_intrinsic, factor=0.5):
    """Front and back cameras are stored at twice the resolution of the side cameras and are halved
    at load time; their projection matrices are scaled with them (reference :358-372)."""
    s = np.eye(4)
    s[0, 0] *= factor
    s[1, 1] *= factor
    out_l2i, out_k = [], []
    for name, l2i, k in zip(img_filenames, lidar2img, cam_intrinsic):
        if name.split("/")[-2] in ("camera_front", "camera_back"):
            out_l2i.append(s @ l2i)
            out_k.append(s @ k)
        else:
            out_l2i.append(l2i)
            out_k.append(k)
    return out_l2i, out_k


def scale_lidar2img(lidar2img, scale):
    """``RandomScaleImageMultiViewImage`` (pipelines/transform_3d.py:295-330): fx, fy, cx, cy scale with the image."""
    s = np.eye(4)
    s[0, 0] *= scale
    s[1, 1] *= scale
    return [s @ m for m in lidar2img]
