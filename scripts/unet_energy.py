#!/usr/bin/env python3
"""Energy per launch class of one UNet forward (batch 16): each class's launches are replayed back to back for about a second with
the package energy counter (rocm-smi --showenergycounter) read before and after.  The forward runs against the 1400 W package cap, so
joules per class -- not microseconds -- say where a change pays (profiles/r02_notes.md sections 11, 16)."""
import sys, os, re, subprocess, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import weights
from coma_amd.sd.unet import HipUNet2DConditionModel
dev = "cuda:0"
B = 16


def energy_uj():
    out = subprocess.run(["rocm-smi", "--showenergycounter"], capture_output=True, text=True).stdout
    m = re.search(r"Accumulated Energy \(uJ\): ([\d.]+)", out)
    return float(m.group(1)) if m else float("nan")


state = weights.random_state(weights.unet_shapes(), seed=0, device=dev)
unet = HipUNet2DConditionModel(state, batch=B, height=64, width=64, device=dev, use_graph=True)
g = torch.Generator(device=dev).manual_seed(0)
unet.set_context(torch.randn(B, 77, 768, generator=g, device=dev))
unet.x_in.copy_(torch.randn(unet.x_in.shape, generator=g, device=dev).half())
unet.timesteps.fill_(961.0)
unet.forward_static(); torch.cuda.synchronize()


def cat_of(tag):
    m = re.match(r"gemm M=(\d+) N=(\d+) K=(\d+) taps=(\d) z=(\d+)", tag)
    if m:
        M, N, K, taps, z = map(int, m.groups())
        if taps == 9:
            return f"conv3x3 M={M}"
        if z > 1:
            return "batched V^T projection"
        if N >= 2 * K and N >= 2560:
            return "GEGLU ff1"
        if K <= 1280 and N <= 1280:
            return "1x1 linears K<=1280"
        return "ff2 and skip 1x1 (K>1280)"
    return (tag.split() or ["other"])[0].split("(")[0]


groups = collections.OrderedDict()
for fn, (tag, fl) in zip(unet.g.launches, unet.g.tags):
    if fn is None:
        continue
    groups.setdefault(cat_of(tag), []).append(fn)


def measure(fns, seconds=1.0):
    for fn in fns:
        fn()
    torch.cuda.synchronize()
    e0 = energy_uj(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(3):
            for fn in fns:
                fn()
        torch.cuda.synchronize(); n += 3
    t1 = time.perf_counter(); e1 = energy_uj()
    return (e1 - e0) * 1e-6 / n, (t1 - t0) / n


idle0 = energy_uj(); time.sleep(1.0); idle_w = (energy_uj() - idle0) * 1e-6
print(f"idle power {idle_w:.0f} W")
fj, ft = measure([f for f in unet.g.launches if f is not None], 2.0)
print(f"whole forward (eager): {ft * 1e3:.2f} ms, {fj:.2f} J -> {fj / ft:.0f} W")
rows = []
for cat, fns in groups.items():
    j, t = measure(fns)
    rows.append((j, t, cat, len(fns)))
tj, tt = sum(r[0] for r in rows), sum(r[1] for r in rows)
for j, t, cat, n in sorted(rows, reverse=True):
    print(f"{j:6.2f} J {100 * j / tj:5.1f}%  {t * 1e3:6.2f} ms {100 * t / tt:5.1f}%  {j / t:5.0f} W  n={n:3d}  {cat}")
print(f"sum over classes: {tj:.2f} J, {tt * 1e3:.2f} ms")
