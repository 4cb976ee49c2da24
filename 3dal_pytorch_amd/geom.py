"""Rotated-box geometry shared by the crop extraction (crops.py) and the training labels (prep.py).

The per-point work — det3d's `points_in_rbbox` (det3d/core/bbox/box_np_ops.py:641-647, inner loop
det3d/core/bbox/geometry.py:240-275) — runs in lib3dal_hip.so. What stays on the host is the O(#boxes) set-up
the reference does before that loop: the six face equations of every box (`center_to_corner_box3d`
box_np_ops.py:241-263, `corner_to_surfaces_3d` :650-671, `surface_equ_3d_jitv2` geometry.py:351-377), computed
here with the same NumPy calls in the boxes' own dtype. They must be NumPy's: the sin/cos of a float32 yaw come
from NumPy's SIMD routines, which device libm does not reproduce bit for bit, and a last-bit change in a face
equation flips points lying next to that face.
"""
import numpy as np
import torch

from . import _hip

# unit-cube corners in det3d's order (corners_nd, box_np_ops.py:55-84) and, per face, the three corners its
# plane equation is built from (the first three of each row of corner_to_surfaces_3d)
_CORNERS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]])
_FACES = np.array([[0, 1, 2], [7, 6, 5], [0, 3, 7], [1, 5, 6], [0, 4, 5], [3, 2, 6]])


def box_planes(rbbox):
    """rbbox (K,7) [x,y,z,l,w,h,yaw], float32 or float64 -> (K,6,4) rows [nx,ny,nz,d] in the same dtype;
    a point is inside when n.p + d < 0 for all six rows."""
    rbbox = np.asarray(rbbox)
    dt = rbbox.dtype
    k = rbbox.shape[0]
    if k == 0:
        return np.zeros((0, 6, 4), dt)
    dims, ang = rbbox[:, 3:6], rbbox[:, -1]
    unit = _CORNERS.astype(dt) - np.array((0.5, 0.5, 0.5), dtype=dt)
    corners = dims.reshape([-1, 1, 3]) * unit.reshape([1, 8, 3])
    s, c = np.sin(ang), np.cos(ang)
    one, zero = np.ones_like(c), np.zeros_like(c)
    corners = np.einsum("aij,jka->aik", corners, np.stack([[c, -s, zero], [s, c, zero], [zero, zero, one]]))
    corners += rbbox[:, :3].reshape([-1, 1, 3])
    p0, p1, p2 = corners[:, _FACES[:, 0]], corners[:, _FACES[:, 1]], corners[:, _FACES[:, 2]]
    a, b = p0 - p1, p1 - p2
    out = np.empty((k, 6, 4), dt)
    out[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    out[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    out[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    out[..., 3] = -p0[..., 0] * out[..., 0] - p0[..., 1] * out[..., 1] - p0[..., 2] * out[..., 2]
    return out


def cull_spheres(rbbox):
    """(K,4) float32 [cx,cy,cz,r^2]: a ball that contains each box with a margin (1 mm + 1e-3 of the radius + 1e-5
    of the largest centre coordinate) far above the fp32 rounding of points and face equations; the crop kernels
    skip the exact test for points outside it. Non-finite boxes get r^2 = inf (no cull)."""
    b = np.asarray(rbbox, dtype=np.float64)
    r = 0.5 * np.sqrt((b[:, 3:6] ** 2).sum(1))
    r = r * 1.001 + 1e-3 + 1e-5 * np.abs(b[:, :3]).max(1, initial=0.0)
    out = np.concatenate([b[:, :3], (r * r)[:, None]], 1)
    out[~np.isfinite(out).all(1), 3] = np.inf
    return np.ascontiguousarray(np.nan_to_num(out, nan=0.0, posinf=np.inf, neginf=0.0), dtype=np.float32)


def planes_to_device(planes, device):
    """face equations are handed to the library as float64 (float32 values are exact in it)"""
    return torch.from_numpy(np.ascontiguousarray(planes, dtype=np.float64)).to(device)


def points_in_rbbox(points, rbbox):
    """det3d's points_in_rbbox on the device. points: CUDA tensor (P,>=3) float32 or float64 (row-contiguous);
    rbbox: NumPy (K,7) float32/float64. Returns a (P,K) bool CUDA tensor. Arithmetic follows NumPy's promotion:
    float32 only when points and boxes both are."""
    if not (torch.is_tensor(points) and points.is_cuda):
        raise RuntimeError("points_in_rbbox: points must be a CUDA tensor (the HIP path has no CPU fallback)")
    if points.dtype not in (torch.float32, torch.float64) or points.dim() != 2 or points.shape[1] < 3 or \
            points.stride(1) != 1:
        raise ValueError("points_in_rbbox: points must be (P,>=3) float32/float64 with unit stride along the row")
    rbbox = np.asarray(rbbox)
    planes = planes_to_device(box_planes(rbbox), points.device)
    p, k = points.shape[0], rbbox.shape[0]
    inside = torch.empty((p, k), dtype=torch.uint8, device=points.device)
    f64 = points.dtype == torch.float64
    f32_math = (not f64) and rbbox.dtype == np.float32
    _hip.check(_hip.lib().dal3_points_in_boxes(_hip.ptr(points), int(f64), p, points.stride(0), _hip.ptr(planes), k,
                                               int(f32_math), _hip.ptr(inside), _hip.stream()))
    return inside.bool()
