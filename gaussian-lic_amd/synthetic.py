"""Seeded synthetic scenes for parity tests and the bench.  CPU torch, fp32.

Raw (pre-activation) parameters follow GaussianModel's layout (src/gaussian.cpp:212-304):
xyz [P,3], scaling = log sigma [P,3], rotation [P,4] (r,x,y,z, unnormalised), opacity = logit [P,1],
features_dc [P,1,3], features_rest [P,15,3] (or [P,0,3] at degree 0).  `activate` applies the activations
render() applies before the rasterizer (src/rasterizer/renderer.cpp:57-63, src/gaussian.cpp:147-175).
"""
import math

import torch

SH_C0 = 0.28209479177387814


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def random_scene(P, W, H, sh_degree=3, seed=0):
    """`random` scene: wide spread of sizes/opacities, 2 % of points behind the near plane (cull path)."""
    g = _gen(seed)
    fx = fy = 0.675 * W
    cx, cy = 0.4857 * W, 0.5215 * H
    U = lambda n, lo, hi: torch.rand(n, generator=g) * (hi - lo) + lo
    N = lambda *s: torch.randn(*s, generator=g)
    u = U(P, -0.1 * W, 1.1 * W)
    v = U(P, -0.1 * H, 1.1 * H)
    z = U(P, 1.0, 30.0)
    behind = torch.rand(P, generator=g) < 0.02
    zb = U(P, -5.0, 0.2)
    z = torch.where(behind, zb, z)
    xyz = torch.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    zs = z.abs().clamp_min(0.2)
    log_sigma = (torch.log(zs / fx) + math.log(2.0) + 0.6 * N(P)).unsqueeze(1) + 0.5 * N(P, 3)
    rot = N(P, 4)
    opacity = 2.0 * N(P, 1)
    dc = ((torch.rand(P, 1, 3, generator=g) - 0.5) / 0.28209479)
    rest = 0.05 * N(P, 15, 3) if sh_degree > 0 else torch.zeros(P, 0, 3)
    return dict(xyz=xyz.float().contiguous(), scaling=log_sigma.float().contiguous(), rotation=rot.float().contiguous(),
                opacity=opacity.float().contiguous(), features_dc=dc.float().contiguous(),
                features_rest=rest.float().contiguous(), sh_degree=int(sh_degree))


def lidar_scene(P, W, H, sh_degree=3, seed=0):
    """`lidar` scene: the initialisation of GaussianModel::initialize (src/gaussian.cpp:212-304) on
    synthetic LiDAR returns — isotropic scale log(z/f), identity quaternions, opacity 0.1, DC colour only."""
    g = _gen(seed)
    fx = fy = 0.675 * W
    cx, cy = 0.4857 * W, 0.5215 * H
    U = lambda n, lo, hi: torch.rand(n, generator=g) * (hi - lo) + lo
    u, v, z = U(P, 0.0, W), U(P, 0.0, H), U(P, 2.0, 40.0)
    xyz = torch.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    f = 0.5 * (fx + fy)
    log_sigma = torch.log(1.0 * z / f).unsqueeze(1).repeat(1, 3)          # gaussian.cpp:222,235
    rot = torch.zeros(P, 4); rot[:, 0] = 1.0                               # gaussian.cpp:238-239
    opacity = torch.full((P, 1), math.log(0.1 / 0.9))                      # inverse_sigmoid(0.1), gaussian.cpp:240
    rgb = torch.rand(P, 1, 3, generator=g)
    dc = (rgb - 0.5) / SH_C0                                               # RGB2SH, gaussian.h:46-47
    rest = torch.zeros(P, 15 if sh_degree > 0 else 0, 3)
    return dict(xyz=xyz.float().contiguous(), scaling=log_sigma.float().contiguous(), rotation=rot.float().contiguous(),
                opacity=opacity.float().contiguous(), features_dc=dc.float().contiguous(),
                features_rest=rest.float().contiguous(), sh_degree=int(sh_degree))


def _quat_from_matrix(R):
    """(r, x, y, z) of a rotation matrix, float64 (Shepperd's branch on the largest diagonal term)."""
    import numpy as np
    R = np.asarray(R, np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = (0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s)
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = ((R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s)
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = ((R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s)
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = ((R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s)
    return np.array(q, np.float64)


def place_scene(raw, R_wc, t_wc):
    """Moves a scene generated in the identity camera frame rigidly into the frame of the pose (R_wc, t_wc): xyz <- R_wc xyz + t_wc and
    every Gaussian's quaternion <- q(R_wc) * q, computed in double and rounded to fp32 once.  A camera at that pose then sees what the
    identity camera saw of the original scene (up to rounding and the view dependence of the SH colour)."""
    import numpy as np
    R = np.asarray(R_wc, np.float64)
    t = np.asarray(t_wc, np.float64)
    out = dict(raw)
    xyz = raw["xyz"].double().numpy() @ R.T + t
    out["xyz"] = torch.from_numpy(xyz).float().contiguous()
    a = _quat_from_matrix(R)
    b = raw["rotation"].double().numpy()
    ar, ax, ay, az = a
    br, bx, by, bz = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    q = np.stack([ar * br - ax * bx - ay * by - az * bz, ar * bx + ax * br + ay * bz - az * by,
                  ar * by - ax * bz + ay * br + az * bx, ar * bz + ax * by - ay * bx + az * br], 1)
    out["rotation"] = torch.from_numpy(q).float().contiguous()
    return out


def pixel_grad(H, W, seed=1):
    """dL/dimage ~ N(0,1), CHW, for backward-only parity."""
    return torch.randn(3, H, W, generator=_gen(seed)).float().contiguous()


def gt_image(H, W, seed=2):
    """Ground-truth image ~ U(0,1), CHW, for loss runs."""
    return torch.rand(3, H, W, generator=_gen(seed)).float().contiguous()


def activate(raw):
    """Activated tensors the rasterizer consumes (renderer.cpp:57-63): sigmoid / exp / normalize."""
    return dict(means=raw["xyz"], scales=torch.exp(raw["scaling"]),
                rots=torch.nn.functional.normalize(raw["rotation"]),
                opac=torch.sigmoid(raw["opacity"]), dc=raw["features_dc"],
                shs=raw["features_rest"], D=raw["sh_degree"])


def to_numpy(act):
    """numpy view of an activated scene for the CPU oracle front-end (oracle/oracle.py, tests only)."""
    out = {}
    for k, v in act.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else v
    return out
