"""Synthetic nuScenes-/KITTI-shaped inputs (SURVEY.md §8(d) "Synthetic input").

There is no dataset on the box: benchmarks and parity tests run on seeded synthetic
sweeps of the shape the reference consumes -- points (x, y, z, intensity, dt) fp32,
six pinhole cameras with 4x4 lidar2cam + 3x3 intrinsics, camera feature maps N(0,1).
numpy only (host-side data generation; nothing here is on the timed path).
"""
import numpy as np

NUSC_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
NUSC_VOXEL = [0.075, 0.075, 0.2]
KITTI_RANGE = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
KITTI_VOXEL = [0.05, 0.05, 0.1]
NUSC_CAMS = ["CAM_FRONT", "CAM_FRONT_LEFT", "CAM_FRONT_RIGHT", "CAM_BACK", "CAM_BACK_LEFT", "CAM_BACK_RIGHT"]


def nusc_sweep(seed=0, n_beams=32, n_az=1875, sweeps=1, pc_range=NUSC_RANGE):
    """One synthetic LiDAR sweep: 32 beams (elevation -30.67..+10.67 deg) x 1875 azimuth steps
    (= 60 000 returns per sweep), ground plane at sensor height 1.84 m, random obstacles at
    5 + 70*U^1.5 m, 1 % range noise; features (x, y, z, intensity in 0..255, dt).
    Returns float32 [P, 5] clipped to pc_range (about 52 k in-range points per sweep)."""
    rs = np.random.RandomState(seed)
    out = []
    for s in range(sweeps):
        elev = np.deg2rad(np.linspace(-30.67, 10.67, n_beams))[:, None]
        az = (np.arange(n_az)[None, :] + rs.uniform(0, 1, size=(n_beams, 1))) * (2 * np.pi / n_az)
        obstacle = 5.0 + 70.0 * rs.uniform(0, 1, size=(n_beams, n_az)) ** 1.5
        # coherent obstacles: low-pass the range image along azimuth so that surfaces exist
        k = 25
        kern = np.ones(k) / k
        obstacle = np.stack([np.convolve(np.r_[o[-k:], o, o[:k]], kern, mode="same")[k:-k] for o in obstacle])
        with np.errstate(divide="ignore"):
            ground = np.where(elev < 0, 1.84 / np.maximum(-np.sin(elev), 1e-6), np.inf)
        r = np.minimum(ground, obstacle) * (1.0 + 0.01 * rs.standard_normal((n_beams, n_az)))
        x = r * np.cos(elev) * np.cos(az)
        y = r * np.cos(elev) * np.sin(az)
        z = r * np.sin(elev)
        inten = rs.randint(0, 256, size=(n_beams, n_az)).astype(np.float64)
        dt = np.full_like(x, 0.05 * s)
        out.append(np.stack([x, y, z, inten, dt], -1).reshape(-1, 5))
    pts = np.concatenate(out, 0).astype(np.float32)
    m = ((pts[:, 0] >= pc_range[0]) & (pts[:, 0] < pc_range[3]) & (pts[:, 1] >= pc_range[1]) &
         (pts[:, 1] < pc_range[4]) & (pts[:, 2] >= pc_range[2]) & (pts[:, 2] < pc_range[5]))
    pts = pts[m]
    rs.shuffle(pts)  # the reference pipelines shuffle points before voxelisation
    return np.ascontiguousarray(pts)


def kitti_sweep(seed=0, n_beams=64, n_az=1900, pc_range=KITTI_RANGE):
    """64-beam sweep restricted to the front +-40 deg FOV (about 18-20 k points), 4 features."""
    rs = np.random.RandomState(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))[:, None]
    az = np.deg2rad(np.linspace(-45, 45, n_az // 4))[None, :] + rs.uniform(0, 1e-3, size=(n_beams, 1))
    obstacle = 5.0 + 65.0 * rs.uniform(0, 1, size=(n_beams, az.shape[1])) ** 1.5
    k = 15
    obstacle = np.stack([np.convolve(o, np.ones(k) / k, mode="same") for o in obstacle])
    ground = np.where(elev < 0, 1.73 / np.maximum(-np.sin(elev), 1e-6), np.inf)
    r = np.minimum(ground, obstacle) * (1.0 + 0.01 * rs.standard_normal(obstacle.shape))
    x = r * np.cos(elev) * np.cos(az)
    y = r * np.cos(elev) * np.sin(az)
    z = r * np.sin(elev)
    inten = rs.uniform(0, 1, size=x.shape)
    pts = np.stack([x, y, z, inten], -1).reshape(-1, 4).astype(np.float32)
    m = ((pts[:, 0] >= pc_range[0]) & (pts[:, 0] < pc_range[3]) & (pts[:, 1] >= pc_range[1]) &
         (pts[:, 1] < pc_range[4]) & (pts[:, 2] >= pc_range[2]) & (pts[:, 2] < pc_range[5]))
    pts = pts[m]
    rs.shuffle(pts)
    return np.ascontiguousarray(pts)


def nusc_cameras(image_hw=(900, 1600), focal=1266.0, yaw_offset_deg=0.0):
    """Six pinhole cameras at 60 deg yaw spacing around the LiDAR (z up, x forward).
    Returns {cam: (lidar2cam 4x4 f32, intrinsic 3x3 f32)} in nuScenes order
    (front, front-left, front-right, back, back-left, back-right).  `yaw_offset_deg` turns the whole rig:
    with 0 the front/back cameras are axis-aligned and voxel corners (a regular lattice) project exactly onto
    pixel boundaries, where 1-ulp differences between implementations flip the reference's truncations."""
    H, W = image_hw
    yaws = {"CAM_FRONT": 0.0, "CAM_FRONT_LEFT": 60.0, "CAM_FRONT_RIGHT": -60.0,
            "CAM_BACK": 180.0, "CAM_BACK_LEFT": 120.0, "CAM_BACK_RIGHT": -120.0}
    cams = {}
    # camera frame: x right, y down, z forward.  For yaw 0 the camera looks along lidar +x.
    base = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    for name in NUSC_CAMS:
        a = np.deg2rad(yaws[name] + yaw_offset_deg)
        rz = np.array([[np.cos(a), np.sin(a), 0.0], [-np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])  # lidar->yawed
        R = base @ rz
        t_cam_in_lidar = np.array([0.3 * np.cos(a), 0.3 * np.sin(a), -0.3])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ t_cam_in_lidar
        Kmat = np.array([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0], [0.0, 0.0, 1.0]])
        cams[name] = (T.astype(np.float32), Kmat.astype(np.float32))
    return cams


def camera_features(n_img, channels=256, hw=(150, 267), seed=1234):
    """Stand-in for the frozen 2D backbone output: N(0,1) fp32 [n_img, C, h, w]."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((n_img, channels, hw[0], hw[1])).astype(np.float32)


def centerhead_targets(batch, num_classes, hw=(180, 180), max_objs=500, seed=0, objs_per_task=(5, 40)):
    """Assigner outputs for `CenterHead.loss` in the reference's layout (CP/det3d/datasets/pipelines/preprocess.py
    `AssignLabel`): per task hm [B, C, H, W] f32 with one Gaussian (radius 2, peak 1) per object, ind [B, M] i64 flat
    centre pixel, mask [B, M] u8, cat [B, M] i64, anno_box [B, M, 10] f32 (sub-pixel offset 2, height 1, log size 3,
    velocity 2, sin / cos of the yaw 2).  Object positions, classes and boxes are random; numpy only."""
    H, W = hw
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[-2:3, -2:3]
    blob = np.exp(-(xx * xx + yy * yy) / (2.0 * (5.0 / 6.0) ** 2)).astype(np.float32)      # sigma = diameter / 6
    ex = dict(hm=[], ind=[], mask=[], cat=[], anno_box=[])
    for nc in num_classes:
        hm = np.zeros((batch, nc, H, W), np.float32)
        ind = np.zeros((batch, max_objs), np.int64)
        mask = np.zeros((batch, max_objs), np.uint8)
        cat = np.zeros((batch, max_objs), np.int64)
        box = np.zeros((batch, max_objs, 10), np.float32)
        for b in range(batch):
            n = int(rs.randint(objs_per_task[0], objs_per_task[1] + 1))
            cx, cy = rs.uniform(2, W - 3, n), rs.uniform(2, H - 3, n)
            ix, iy = cx.astype(np.int64), cy.astype(np.int64)
            cls = rs.randint(0, nc, n)
            for m in range(n):
                win = hm[b, cls[m], iy[m] - 2:iy[m] + 3, ix[m] - 2:ix[m] + 3]
                np.maximum(win, blob, out=win)
            yaw = rs.uniform(-np.pi, np.pi, n)
            ind[b, :n], mask[b, :n], cat[b, :n] = iy * W + ix, 1, cls
            box[b, :n] = np.stack([cx - ix, cy - iy, rs.uniform(-2, 1, n), np.log(rs.uniform(0.5, 5, n)),
                                   np.log(rs.uniform(0.5, 12, n)), np.log(rs.uniform(0.5, 4, n)), rs.normal(0, 2, n),
                                   rs.normal(0, 2, n), np.sin(yaw), np.cos(yaw)], 1).astype(np.float32)
        ex["hm"].append(hm), ex["ind"].append(ind), ex["mask"].append(mask), ex["cat"].append(cat)
        ex["anno_box"].append(box)
    return ex


def nusc_gt_boxes(seed=0, objs=(15, 60), num_classes=10, pc_range=NUSC_RANGE):
    """Ground truth of one nuScenes-shaped sample as mmdet3d hands it to the head: boxes [G, 9] f32
    (x, y, z_bottom, w, l, h, yaw, vx, vy) inside the BEV range and a class id per box [G] i64."""
    rs = np.random.RandomState(90000 + seed)
    n = int(rs.randint(objs[0], objs[1] + 1))
    xy = rs.uniform(0.92 * pc_range[0], 0.92 * pc_range[3], (n, 2))
    cls = rs.randint(0, num_classes, n)
    base = np.array([[1.95, 4.6, 1.7], [2.5, 6.9, 2.8], [2.9, 6.4, 3.2], [2.9, 11.0, 3.5], [2.9, 12.3, 3.9], [2.5, 0.5, 1.0],
                     [0.8, 2.1, 1.5], [0.6, 1.7, 1.3], [0.7, 0.7, 1.8], [0.4, 0.4, 1.1]])[cls % 10]
    dims = base * np.exp(rs.normal(0, 0.12, (n, 3)))
    boxes = np.concatenate([xy, rs.uniform(-2.5, -0.5, (n, 1)), dims, rs.uniform(-np.pi, np.pi, (n, 1)),
                            rs.normal(0, 1.5, (n, 2))], 1).astype(np.float32)
    return boxes, cls.astype(np.int64)
