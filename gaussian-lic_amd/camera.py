"""Camera input contract of the hot path — mirrors Camera::setIntrinsic/setPose/setWorldViewTransform/
setProjectionMatrix of the reference (src/camera.h:38-110).  Pure numpy, CPU; produces exactly the
float[16] layouts the kernels read (element (row r, col c) at [4c+r], auxiliary.h:70-89)."""
import math

import numpy as np


class Camera:
    def __init__(self, width, height, fx, fy, cx, cy, R_wc=None, t_wc=None, znear=0.01, zfar=100.0):
        self.image_width, self.image_height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = (np.float32(v) for v in (fx, fy, cx, cy))
        self.znear, self.zfar = np.float32(znear), np.float32(zfar)
        # camera.h:49-50 (double math, stored in float members)
        self.FoVx = np.float32(2.0 * math.atan(width / (2.0 * float(fx))))
        self.FoVy = np.float32(2.0 * math.atan(height / (2.0 * float(fy))))
        self.set_pose(np.eye(3) if R_wc is None else R_wc, np.zeros(3) if t_wc is None else t_wc)

    def set_pose(self, R_wc, t_wc):
        R_wc = np.asarray(R_wc, np.float64)
        t_wc = np.asarray(t_wc, np.float64)
        self.R_wc, self.t_wc = R_wc.copy(), t_wc.copy()
        R_cw = R_wc.T
        t_cw = -R_wc.T @ t_wc
        Rt = np.zeros((4, 4), np.float32)                      # camera.h:70-77 (trans_=0, scale_=1)
        Rt[:3, :3] = R_cw.astype(np.float32)
        Rt[:3, 3] = t_cw.astype(np.float32)
        Rt[3, 3] = 1.0
        Rt = np.linalg.inv(np.linalg.inv(Rt)).astype(np.float32)  # camera.h:79-84 inverts twice
        self.world_view_transform = np.ascontiguousarray(Rt.T)    # camera.h:86 (.transpose(0,1))
        P = np.zeros((4, 4), np.float32)                        # camera.h:89-109
        W, H = np.float32(self.image_width), np.float32(self.image_height)
        P[0, 0] = 1.0 / math.tan(float(self.FoVx) / 2)
        P[1, 1] = 1.0 / math.tan(float(self.FoVy) / 2)
        P[0, 2] = (2 * self.cx - W) / W
        P[1, 2] = (2 * self.cy - H) / H
        P[3, 2] = 1.0
        P[2, 2] = self.zfar / (self.zfar - self.znear)
        P[2, 3] = -(self.zfar * self.znear) / (self.zfar - self.znear)
        self.projection_matrix = np.ascontiguousarray(P.T)
        self.full_proj_transform = np.ascontiguousarray(
            (self.world_view_transform @ self.projection_matrix).astype(np.float32))   # camera.h:60
        self.camera_center = np.ascontiguousarray(
            np.linalg.inv(self.world_view_transform)[3, :3].astype(np.float32))         # camera.h:61
        w, h = float(self.image_width), float(self.image_height)
        fx, fy, cx, cy = float(self.fx), float(self.fy), float(self.cx), float(self.cy)
        self.limx_neg = np.float32(-0.15 * w / fx - cx / fx)   # camera.h:63-66
        self.limx_pos = np.float32(1.15 * w / fx - cx / fx)
        self.limy_neg = np.float32(-0.15 * h / fy - cy / fy)
        self.limy_pos = np.float32(1.15 * h / fy - cy / fy)
        # renderer.cpp:31-32: std::tan(FoV * 0.5f) in float
        self.tanfovx = np.float32(np.tan(np.float32(self.FoVx * np.float32(0.5))))
        self.tanfovy = np.float32(np.tan(np.float32(self.FoVy * np.float32(0.5))))

    # ---- camera pose as an optimisation variable (the "cam" of the north-star; the reference has no counterpart: rasterizer.cpp:171-182) --------
    def pose_gradient(self, dL_dviewmatrix, dL_dprojmatrix, dL_dcampos):
        """Chains the three camera gradients of gslic_rasterize_backward_camera (element order of the inputs: float[16] with (r, c) at [4c + r])
        to the six coordinates of a LEFT pose increment  T_cw <- exp(xi^) T_cw,  xi = (rho, phi) in se(3)  (rho: translation, phi: rotation,
        both in the camera frame).  With V = [R | t] the world-to-camera matrix, the full projection P V and the centre c = -R^T t:
            G = dL/dV + P^T dL/d(PV)            (top three rows)
            G_R = G[:, :3] - t g_c^T,   G_t = G[:, 3] - R g_c
            dL/drho = G_t,   dL/dphi = vee(M - M^T) + t x G_t,   M = G_R R^T
        Returns float64 [6] = (dL/drho, dL/dphi)."""
        f64 = lambda a, shape: np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, np.float64).reshape(shape)
        Gv, Gp, gc = f64(dL_dviewmatrix, (4, 4)).T, f64(dL_dprojmatrix, (4, 4)).T, f64(dL_dcampos, (3,))   # stored transposed -> math matrices
        V = np.asarray(self.world_view_transform, np.float64).T
        Pm = np.asarray(self.projection_matrix, np.float64).T
        R, t = V[:3, :3], V[:3, 3]
        G = (Gv + Pm.T @ Gp)[:3, :]
        G_R = G[:, :3] - np.outer(t, gc)
        G_t = G[:, 3] - R @ gc
        M = G_R @ R.T
        dphi = np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]]) + np.cross(t, G_t)
        return np.concatenate([G_t, dphi])

    def apply_pose_increment(self, xi):
        """T_cw <- exp(xi^) T_cw (xi = (rho, phi) as in pose_gradient): every derived matrix is recomputed the way camera.h:70-110 does."""
        E = se3_exp(xi)
        R_cw, t_cw = self.R_wc.T, -self.R_wc.T @ self.t_wc
        R_new, t_new = E[:3, :3] @ R_cw, E[:3, :3] @ t_cw + E[:3, 3]
        self.set_pose(R_new.T, -R_new.T @ t_new)
        return self

    def to_device(self, device):
        """Attach device copies of the three tensors the rasterizer reads (camera.h:86,109,60-61 keep them on CUDA)."""
        import torch
        self.d_world_view_transform = torch.from_numpy(self.world_view_transform).to(device)
        self.d_full_proj_transform = torch.from_numpy(self.full_proj_transform).to(device)
        self.d_camera_center = torch.from_numpy(self.camera_center).to(device)
        return self

    def as_dict(self):
        """Plain dict consumed by the oracle front-end and the HIP front-end alike."""
        return dict(W=self.image_width, H=self.image_height,
                    view=self.world_view_transform.reshape(16).copy(),
                    proj=self.full_proj_transform.reshape(16).copy(),
                    campos=self.camera_center.copy(),
                    tanfovx=float(self.tanfovx), tanfovy=float(self.tanfovy),
                    limx_neg=float(self.limx_neg), limx_pos=float(self.limx_pos),
                    limy_neg=float(self.limy_neg), limy_pos=float(self.limy_pos))


def se3_exp(xi):
    """exp of (rho, phi) in se(3) as a 4x4 matrix (Rodrigues + the left Jacobian for the translation), float64."""
    xi = np.asarray(xi, np.float64)
    rho, phi = xi[:3], xi[3:]
    th = float(np.linalg.norm(phi))
    K = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    if th < 1e-8:
        Rm, Vm = np.eye(3) + K + 0.5 * K @ K, np.eye(3) + 0.5 * K + K @ K / 6.0
    else:
        a, b, c = math.sin(th) / th, (1.0 - math.cos(th)) / th ** 2, (th - math.sin(th)) / th ** 3
        Rm, Vm = np.eye(3) + a * K + b * K @ K, np.eye(3) + b * K + c * K @ K
    E = np.eye(4)
    E[:3, :3], E[:3, 3] = Rm, Vm @ rho
    return E


def rotation_ypr(yaw_deg, pitch_deg, roll_deg):
    """R_wc = R_y(yaw) R_x(pitch) R_z(roll) (camera convention of the scenes: +x right, +y down, +z forward), float64."""
    y, p, r = (math.radians(float(v)) for v in (yaw_deg, pitch_deg, roll_deg))
    Ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


# General SE(3) poses of the parity suites (pitch AND roll >= 20 deg, translation on all axes, so that the W matrix of
# forward.cu:101-104 / backward.cu:185 is dense and campos != 0).  "place": the scene (generated in the identity camera frame) is moved rigidly
# into this pose's frame, so the camera still sees all of it; without it the camera looks at the identity-frame scene from the side: a large part
# of it leaves the image, and Gaussians beyond the 15 % margins have their cov2D Jacobian clamped (forward.cu:91-94, backward.cu:177-178).
SE3_POSES = {
    "se3_a": dict(ypr=(25.0, -22.0, 31.0), t=(0.8, -0.5, 0.6), place=True),
    "se3_b": dict(ypr=(-33.0, 24.0, -27.0), t=(-1.2, 0.7, -0.9), place=True),
    "se3_c": dict(ypr=(21.0, -20.0, 23.0), t=(0.4, 0.3, -0.5), place=False),
    "se3_d": dict(ypr=(-12.0, 26.0, -40.0), t=(-0.3, -0.6, 0.2), place=False),
}


def resolve_view(view):
    """view spec -> (R_wc, t_wc, place) or None for the identity pose.  Accepted: None; k in 0..7 (SURVEY.md section 8d's config-4 views);
    a name of SE3_POSES; a dict(ypr=(yaw, pitch, roll) degrees, t=(x, y, z), place=bool)."""
    if view is None:
        return None
    if isinstance(view, str):
        view = SE3_POSES[view]
    if isinstance(view, dict):
        return rotation_ypr(*view["ypr"]), np.asarray(view["t"], np.float64), bool(view.get("place", False))
    a = math.radians((int(view) - 3.5) * 4.0)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    return R, np.array([(int(view) - 3.5) * 0.25, 0.0, 0.0]), False


def synthetic_camera(W, H, view_index=None):
    """SURVEY.md §8d camera: fx = fy = 0.675 W, cx = 0.4857 W, cy = 0.5215 H (config/fastlivo.yaml:1-6 ratios).
    view_index None -> identity pose; k in 0..7 -> yaw (k-3.5)*4 deg about +y, translation x = (k-3.5)*0.25 m; a name of SE3_POSES or a
    dict(ypr, t) -> that general pose (resolve_view)."""
    fx = fy = 0.675 * W
    cx, cy = 0.4857 * W, 0.5215 * H
    pose = resolve_view(view_index)
    if pose is None:
        return Camera(W, H, fx, fy, cx, cy)
    return Camera(W, H, fx, fy, cx, cy, pose[0], pose[1])
