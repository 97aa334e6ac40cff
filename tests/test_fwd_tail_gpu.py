"""-m gpu: the forward blend's tail (render_fwd_tail_kernel, csrc/render.hip) — the last tiles of the launch blended by FOUR waves, one 8x8 quadrant
each, instead of two — changes which wave owns a pixel and nothing else: image, final_T, the checkpoints / decision masks the backward reads (seen
through the gradients and the Adam state after a fused step) must be the same BITS for every tail fraction, in both arithmetic modes.  The fraction
is read once per process (GSLIC_FWD_TAIL4), so every setting runs in a process of its own and reports digests.  (The comparison with the reference's
own kernels on the default fraction is every other -m gpu suite; with half of every scene's tiles on four waves: profiles/r06y_fuzz_150_tail_half.txt.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
import gaussian_lic_amd
from gaussian_lic_amd import trainer, _lib
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.rasterizer import render
from gaussian_lic_amd.synthetic import random_scene, gt_image
P, W, H, strict = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
_lib.lib().gslic_set_math_mode(strict)
dev = torch.device("cuda:0")
def dig(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()
model = trainer.GaussianModel(random_scene(P, W, H, 3, 11), dev, order="morton")
model.training_setup()
cam = synthetic_camera(W, H).to_device(dev); gt = gt_image(H, W, seed=3).to(dev); bg = torch.zeros(3, device=dev)
out = {}
with torch.no_grad():
    image, _, _, visible, radii = render(cam, model, bg)
out["image"], out["radii"] = dig(image), dig(radii)
for _ in range(3):
    trainer.training_step_fused(model, cam, gt, bg)
torch.cuda.synchronize()
for n in model.NAMES:
    out["p_" + n] = dig(getattr(model, n)); out["m_" + n] = dig(model._m[n][:model.P]); out["v_" + n] = dig(model._v[n][:model.P])
print("DIGESTS " + json.dumps(out))
""" % ROOT


def _run(tail, P, W, H, strict):
    env = dict(os.environ)
    if tail is None:
        env.pop("GSLIC_FWD_TAIL4", None)
    else:
        env["GSLIC_FWD_TAIL4"] = str(tail)
    r = subprocess.run([sys.executable, "-c", PROBE, str(P), str(W), str(H), str(strict)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")][-1]
    return json.loads(line[len("DIGESTS "):])


@pytest.mark.parametrize("P,W,H", [(120000, 640, 368), (40000, 200, 120)])
@pytest.mark.parametrize("strict", [1, 0])
def test_tail_fraction_changes_no_bit(P, W, H, strict):
    ref = _run(0, P, W, H, strict)                 # two waves per tile everywhere (render_fwd_kernel<*, 2>)
    for tail in (None, 0.5, 1.0):                  # the default fraction; half of the tiles; every tile on four waves (through the tail kernel)
        got = _run(tail, P, W, H, strict)
        diff = [k for k in ref if ref[k] != got[k]]
        assert not diff, f"GSLIC_FWD_TAIL4={tail}: {diff} differ from the two-wave launch"
