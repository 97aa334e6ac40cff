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


def synthetic_camera(W, H, view_index=None):
    """SURVEY.md §8d camera: fx = fy = 0.675 W, cx = 0.4857 W, cy = 0.5215 H (config/fastlivo.yaml:1-6 ratios).
    view_index None -> identity pose; k in 0..7 -> yaw (k-3.5)*4 deg about +y, translation x = (k-3.5)*0.25 m."""
    fx = fy = 0.675 * W
    cx, cy = 0.4857 * W, 0.5215 * H
    if view_index is None:
        return Camera(W, H, fx, fy, cx, cy)
    a = math.radians((view_index - 3.5) * 4.0)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    t = np.array([(view_index - 3.5) * 0.25, 0.0, 0.0])
    return Camera(W, H, fx, fy, cx, cy, R, t)
