"""Deterministic stand-ins for the two third-party segmentation models behind the mask plug-ins (TEST INFRASTRUCTURE):
PointRend (detectron2 `DefaultPredictor`) and SAM (`SamPredictor`).  They let the predictor LOGIC -- category filter, mask
merging, box policies, asset exclusion -- be exercised and pinned against the reference classes without the weights.

  fake_pointrend(image) -> masks bool [N,H,W], scores f32 [N], classes i64 [N]
      one instance per saturated colour blob: red blobs are class 0 ("person"), blue blobs class 56; score = blob area ratio
  FakeSam.set_image(image); FakeSam.predict(box=, multimask_output=) -> (masks [K,H,W] bool, scores [K], logits)
      masks = the bright pixels (green channel > 100) inside the box, (K = 3 shrunken variants with multimask_output)
"""
import numpy as np


def _blobs(chan_mask):
    """connected components by row-run flood (tiny images, test only) -> list of bool masks, largest first"""
    H, W = chan_mask.shape
    seen = np.zeros_like(chan_mask, dtype=bool)
    out = []
    for y in range(H):
        for x in range(W):
            if chan_mask[y, x] and not seen[y, x]:
                stack, m = [(y, x)], np.zeros_like(chan_mask, dtype=bool)
                seen[y, x] = True
                while stack:
                    cy, cx = stack.pop()
                    m[cy, cx] = True
                    for ny, nx in ((cy + 1, cx), (cy - 1, cx), (cy, cx + 1), (cy, cx - 1)):
                        if 0 <= ny < H and 0 <= nx < W and chan_mask[ny, nx] and not seen[ny, nx]:
                            seen[ny, nx] = True
                            stack.append((ny, nx))
                out.append(m)
    return sorted(out, key=lambda m: -int(m.sum()))


def fake_pointrend(image):
    img = np.asarray(image)
    red = (img[..., 0] > 200) & (img[..., 1] < 120) & (img[..., 2] < 80)
    blue = (img[..., 2] > 200) & (img[..., 0] < 80)
    masks, scores, classes = [], [], []
    for cls, chan in ((0, red), (56, blue)):
        for m in _blobs(chan):
            masks.append(m)
            scores.append(m.sum() / m.size + 0.5)
            classes.append(cls)
    H, W = img.shape[:2]
    if not masks:
        return np.zeros((0, H, W), bool), np.zeros((0,), np.float32), np.zeros((0,), np.int64)
    return np.stack(masks), np.asarray(scores, np.float32), np.asarray(classes, np.int64)


class FakeSam:
    def __init__(self):
        self.image = None

    def set_image(self, image):
        self.image = np.asarray(image)

    def predict(self, box=None, multimask_output=False, **_):
        assert self.image is not None, "set_image first"
        x0, y0, x1, y1 = [int(v) for v in box]
        bright = self.image[..., 1] > 100
        inside = np.zeros_like(bright)
        inside[y0:y1, x0:x1] = True
        base = bright & inside
        if not multimask_output:
            return base[None], np.array([0.9], np.float32), None
        ms = [base, base & np.roll(base, 1, 0), base & np.roll(base, 1, 1)]
        return np.stack(ms), np.array([0.5, 0.8, 0.7], np.float32), None


def scene(seed, H=40, W=48, person=True, n_person=1, asset=True):
    """An RGB test frame: red 'person' blobs with a green-bright interior, a blue 'chair', a green-bright 'asset' patch."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 60, size=(H, W, 3)).astype(np.uint8)
    amask = np.zeros((H, W), np.uint8)
    if asset:
        ay, ax = int(rng.integers(20, 28)), int(rng.integers(4, 30))
        img[ay:ay + 10, ax:ax + 14, 1] = 180
        amask[ay:ay + 10, ax:ax + 14] = 1
    if person:
        for k in range(n_person):
            py, px = int(rng.integers(2, 14)), int(rng.integers(2 + 22 * k, 8 + 22 * k))
            h, w = int(rng.integers(8, 14)), int(rng.integers(5, 10))
            img[py:py + h, px:px + w] = [230, 110, 20]
            img[py + 1:py + h + 6, px + 1:px + w - 1, 1] = np.maximum(img[py + 1:py + h + 6, px + 1:px + w - 1, 1], 110)
    by, bx = int(rng.integers(30, 36)), int(rng.integers(30, 40))
    img[by:by + 4, bx:bx + 6] = [10, 40, 240]
    return img, amask
