#!/usr/bin/env python3
"""One UNet graph at batch 16 vs two half-batch graphs replayed concurrently on two streams (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import weights
from coma_amd.sd.unet import HipUNet2DConditionModel
dev = "cuda:0"
state = weights.random_state(weights.unet_shapes(), seed=0, device=dev)


def mk(B):
    u = HipUNet2DConditionModel(state, batch=B, height=64, width=64, device=dev, use_graph=True, cfg_shared_prefix=True)
    g = torch.Generator(device=dev).manual_seed(B)
    u.set_context(torch.randn(B, 77, 768, generator=g, device=dev))
    half = torch.randn(B // 2, 4096, 64, generator=g, device=dev).half()
    u.x_in.copy_(torch.cat([half, half]))
    u.timesteps.fill_(961.0)
    u.forward_static()
    torch.cuda.synchronize()
    return u


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n


u16 = mk(16)
print(f"one graph, batch 16: {timeit(u16.forward_static):.2f} ms")
ua, ub = mk(8), mk(8)
print(f"one graph, batch 8: {timeit(ua.forward_static):.2f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ua.forward_static()
    with torch.cuda.stream(s2):
        ub.forward_static()
    cur.wait_stream(s1); cur.wait_stream(s2)


print(f"two graphs, batch 8 each, two streams: {timeit(dual):.2f} ms")
if len(sys.argv) > 1:
    uq = [mk(4) for _ in range(4)]
    ss = [torch.cuda.Stream() for _ in range(4)]

    def quad():
        cur = torch.cuda.current_stream()
        for s, u in zip(ss, uq):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                u.forward_static()
        for s in ss:
            cur.wait_stream(s)
    print(f"four graphs, batch 4 each: {timeit(quad):.2f} ms")
