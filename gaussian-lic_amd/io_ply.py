"""Map export / import — the wire format of GaussianModel::saveMap (src/gaussian.cpp:306-397, written there through the
vendored tinyply): binary little-endian PLY, one `vertex` element, float32 properties in this order
    x y z | f_dc_0..2 | f_rest_0..(3M-1) | opacity | scale_0..2 | rot_0..3
with f_dc / f_rest stored CHANNEL-major (the [P,K,3] tensors are transposed to [P,3,K] and flattened, :312-313), raw
(pre-activation) opacity / scale / rotation, and the first `skybox_points_num` rows dropped (:310-316).
SURVEY.md §8f row 3: viewer interop and a way to load real scenes into the benchmark."""
import numpy as np
import torch


def _props(M):
    names = ["x", "y", "z"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * M)] + ["opacity"]
    return names + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_map(model, path, skybox_points_num=0):
    """model: anything with xyz, features_dc [P,1,3], features_rest [P,M,3], opacity [P,1], scaling [P,3], rotation [P,4]."""
    s = int(skybox_points_num)
    order = model.original_order() if callable(getattr(model, "original_order", None)) else None   # a model that stores its rows permuted
    g = lambda t: (t.detach() if order is None else t.detach()[order])[s:].float().cpu()
    xyz, dc, rest = g(model.xyz), g(model.features_dc), g(model.features_rest)
    M = rest.shape[1]
    cols = [xyz, dc.transpose(1, 2).flatten(1), rest.transpose(1, 2).flatten(1), g(model.opacity), g(model.scaling), g(model.rotation)]
    data = torch.cat([c.reshape(xyz.shape[0], -1) for c in cols], 1).contiguous().numpy().astype("<f4")
    names = _props(M)
    assert data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {data.shape[0]}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())
    return data.shape[0]


def load_map(path, sh_degree=None):
    """Inverse of save_map -> raw-parameter dict in the layout of gaussian_lic_amd.synthetic (CPU tensors)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        assert f.readline().strip() == b"format binary_little_endian 1.0"
        n, names = 0, []
        while True:
            line = f.readline().decode("ascii").strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[:2] == ["element", "vertex"]:
                n = int(tok[2])
            elif tok[0] == "property":
                assert tok[1] in ("float", "float32"), "saveMap writes float32 properties only"
                names.append(tok[2])
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    col = {nm: i for i, nm in enumerate(names)}
    M = sum(1 for nm in names if nm.startswith("f_rest_")) // 3
    take = lambda keys: torch.from_numpy(np.stack([data[:, col[k]] for k in keys], 1).astype(np.float32)) if keys else torch.zeros(n, 0)
    dc = take([f"f_dc_{i}" for i in range(3)]).reshape(n, 3, 1).transpose(1, 2).contiguous()
    rest = take([f"f_rest_{i}" for i in range(3 * M)]).reshape(n, 3, M).transpose(1, 2).contiguous()
    deg = sh_degree if sh_degree is not None else {0: 0, 3: 1, 8: 2, 15: 3}.get(M, 3)
    return dict(xyz=take(["x", "y", "z"]), scaling=take([f"scale_{i}" for i in range(3)]), rotation=take([f"rot_{i}" for i in range(4)]),
                opacity=take(["opacity"]), features_dc=dc, features_rest=rest, sh_degree=deg)
