"""Seeded synthetic (human, object) samples shared by the golden generator, the parity tests and bench.py."""
import numpy as np


def synth_sample(rng, H, O, obj_pts=None, obj_nrm=None, spread=0.12):
    """Humans scattered around the object so that some pairs touch; unit normals."""
    if obj_pts is None:
        d = rng.normal(size=(O, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        obj_pts = d * 0.2 + np.array([0.0, -0.15, 0.3])
        obj_nrm = d.copy()
    hv = obj_pts[rng.integers(0, O, size=H)] + rng.normal(scale=spread, size=(H, 3))
    hn = rng.normal(size=(H, 3))
    hn /= np.linalg.norm(hn, axis=1, keepdims=True)
    return dict(human_verts=hv, human_normals=hn, obj_verts=obj_pts, obj_normals=obj_nrm)


def make_samples(H, O, S, seed, thres, const_obj=True):
    """S samples; sample 0 carries four pairs placed within a few f32 ulps of the contact threshold."""
    rng = np.random.default_rng(seed)
    s0 = synth_sample(rng, H, O)
    out = []
    for s in range(S):
        smp = synth_sample(rng, H, O, s0["obj_verts"], s0["obj_normals"]) if const_obj else synth_sample(rng, H, O)
        if s == 0:
            for j in range(min(4, H)):
                dirn = rng.normal(size=3)
                dirn /= np.linalg.norm(dirn)
                smp["human_verts"][j] = smp["obj_verts"][j % O] + dirn * np.float64(np.float32(thres)) * (1 + (j - 2) * 6e-8)
        out.append(smp)
    return out


def cfg1_samples(S, seed=0, H=1000, O=180):
    """BASELINE.json config 1 inputs: humans uniform in a body-sized box, object points on
    a sphere r=0.2 m at (0,-0.15,0.3) with outward normals, constant across samples."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(O, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ov, on = d * 0.2 + np.array([0.0, -0.15, 0.3]), d.copy()
    out = []
    for _ in range(S):
        hv = rng.uniform([-0.3, -0.15, -0.85], [0.3, 0.15, 0.85], size=(H, 3))
        hn = rng.normal(size=(H, 3))
        hn /= np.linalg.norm(hn, axis=1, keepdims=True)
        out.append(dict(human_verts=hv, human_normals=hn, obj_verts=ov, obj_normals=on))
    return out
