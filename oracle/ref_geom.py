"""ORACLE — CPU restatement of the reference's points-in-rotated-box test and of the crop extraction built on it
(SURVEY.md 8(f) N1 mask labels and N2 crop extraction).

TEST INFRASTRUCTURE ONLY (same rules as oracle/ref_heads.py): imported by tests/, never by the product.

What it follows
  * `points_in_rbbox`            det3d/core/bbox/box_np_ops.py:641-647
      `center_to_corner_box3d`   box_np_ops.py:241-263 (`corners_nd` :55-84, `rotation_3d_in_axis` :146-178, axis=2)
      `corner_to_surfaces_3d`    box_np_ops.py:650-671
      `surface_equ_3d_jitv2`     det3d/core/bbox/geometry.py:351-377
      `_points_in_convex_polygon_3d_jit`  geometry.py:240-275 (a point is OUTSIDE as soon as one face gives
                                 n.p + d >= 0; a NaN never compares >= 0, so NaN points count as inside)
  * the per-detection part of `_create_pd_detection`, det3d/datasets/waymo/waymo_common.py:105-111 (CenterPoint
    box -> Waymo box), :168-171 (points of the sweep inside the detection, moved to the global frame), :193
    (`transform_box`, :52-65).

Parity pin: the geometry routines are `numba.njit` functions and numba is not in this image, but their bodies
are plain Python; tests/golden/gen_golden.py registers a `numba` module whose decorators return the function
unchanged, loads the reference's own geometry.py / box_np_ops.py / waymo_common.py from /root/reference and runs
them in CPython. numba compiles the same IEEE operation sequence (no fastmath is requested anywhere in those
files), so the fixtures tests/golden/geom_rbbox.npz and crops_extract.npz are outputs of the reference's code;
tests/test_oracle_geom.py checks this file against them bit for bit.

Quirk kept: `rotation_3d_in_axis` turns the corners CLOCKWISE by the yaw (row vectors times [[c,-s],[s,c]]), the
convention of det3d's own [w,l,h,r2] boxes; the reference calls it with Waymo-convention boxes (counter-clockwise
yaw), so the tested region is the box mirrored in yaw. The restatement does what the reference does.

dtype rules kept from NumPy: planes are computed in the boxes' dtype (float32 boxes -> float32 planes); the
per-point expression is evaluated in the promoted type of point and plane (float32 only when both are float32).
"""
import numpy as np

# corner order of corners_nd for ndim=3 after its [0,1,3,2,4,5,7,6] re-ordering, as 0/1 unit coordinates
_UNIT = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]])
# the three corners of each face that the plane equation uses (first three of corner_to_surfaces_3d's four)
_FACE = np.array([[0, 1, 2], [7, 6, 5], [0, 3, 7], [1, 5, 6], [0, 4, 5], [3, 2, 6]])


def box_corners(rbbox):
    """center_to_corner_box3d(rbbox[:, :3], rbbox[:, 3:6], rbbox[:, -1], origin=.5, axis=2) -> (K,8,3)"""
    dims = rbbox[:, 3:6]
    unit = _UNIT.astype(dims.dtype) - np.array((0.5, 0.5, 0.5), dtype=dims.dtype)
    corners = dims.reshape([-1, 1, 3]) * unit.reshape([1, 8, 3])
    ang = rbbox[:, -1]
    s, c = np.sin(ang), np.cos(ang)
    one, zero = np.ones_like(c), np.zeros_like(c)
    rot_t = np.stack([[c, -s, zero], [s, c, zero], [zero, zero, one]])
    corners = np.einsum("aij,jka->aik", corners, rot_t)
    corners += rbbox[:, :3].reshape([-1, 1, 3])
    return corners


def box_planes(rbbox):
    """surface_equ_3d_jitv2 of the six faces: normals (K,6,3) and offsets (K,6) in the boxes' dtype, oriented as the
    reference has them (inside <=> n.p + d < 0 for all six)."""
    corners = box_corners(rbbox)
    p0, p1, p2 = (corners[:, _FACE[:, i], :] for i in range(3))          # (K,6,3)
    a, b = p0 - p1, p1 - p2
    n = np.empty_like(a)
    n[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    n[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    n[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    d = -p0[..., 0] * n[..., 0] - p0[..., 1] * n[..., 1] - p0[..., 2] * n[..., 2]
    return n, d


def points_in_rbbox(points, rbbox):
    """(P,>=3), (K,7) -> (P,K) bool"""
    n, d = box_planes(rbbox)
    p = points[:, :3]
    sign = (p[:, None, None, 0] * n[None, :, :, 0] + p[:, None, None, 1] * n[None, :, :, 1]
            + p[:, None, None, 2] * n[None, :, :, 2] + d[None])                                  # (P,K,6)
    return ~(sign >= 0).any(axis=2)


def transform_box(box, pose):
    """waymo_common.py:52-65"""
    heading = box[..., -1] + np.arctan2(pose[..., 1, 0], pose[..., 0, 0])
    center = np.einsum("...ij,...nj->...ni", pose[..., 0:3, 0:3], box[..., 0:3]) + np.expand_dims(pose[..., 0:3, 3], axis=-2)
    return np.squeeze(np.concatenate([center, box[..., 3:6], heading[..., np.newaxis]], axis=-1))


def waymo_boxes(box3d_lidar):
    """waymo_common.py:105-111: detector boxes [x,y,z,w,l,h,(vx,vy,)r2] -> [x,y,z,l,w,h,r1], r1 = -r2 - pi/2"""
    b = np.array(box3d_lidar, copy=True)
    b[:, -1] = -b[:, -1] - np.pi / 2
    return b[:, [0, 1, 2, 4, 3, 5, -1]]


def extract_crops(lidar_xyz, box3d_lidar, veh_to_global):
    """One frame of _create_pd_detection's track-data extraction. Returns (boxes_lidar (K,7) vehicle frame,
    [global box (7,)] * K, [points (k,3) float64 global frame] * K)."""
    pose = np.reshape(veh_to_global, [4, 4])
    box3d = waymo_boxes(box3d_lidar)
    boxes_g, pts_g = [], []
    for i in range(box3d.shape[0]):
        det = box3d[i]
        inside = points_in_rbbox(lidar_xyz, det[np.newaxis, ...]).reshape([-1])
        o = lidar_xyz[inside].T
        o = pose @ np.concatenate([o, np.ones((1, o.shape[1]))], axis=0)
        pts_g.append(o[:3, :].T)
        boxes_g.append(transform_box(det[np.newaxis, ...], pose))
    return box3d, boxes_g, pts_g
