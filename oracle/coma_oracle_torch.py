"""torch-CPU restatement of the ComA contact / orientation accumulation with the reference's dtype flow, multi-threaded
through ATen -- TEST / BASELINE INFRASTRUCTURE ONLY (nothing under coma_amd/, utils/, src/ imports it).

It exists for one purpose: bench.py's `cpu_baseline` (SURVEY.md 8d "CPU baseline timing plan": the reference-faithful
torch-CPU flow on all host threads), next to the single-threaded NumPy figure of oracle/coma_oracle.py.
Follows /root/reference/utils/coma.py:279-323 (aggregate_single_sample_for_contact), :123-172 (canonicalize_a_wrt_b_to_p),
:102-112 (geodesic_gaussian_scores): f32 distances / canonical normals, f64 sphere bins, f64 scores added in place into f32
grids.  Checked against the NumPy oracle (itself pinned to the reference by G1-G7) in tests/test_oracle_golden.py.
"""
import math

import torch

F32, F64 = torch.float32, torch.float64


def fibonacci_sphere(n):
    i = torch.arange(0, n, dtype=F64) + 0.5
    phi = torch.acos(1 - 2 * i / n)
    theta = math.pi * (1 + 5 ** 0.5) * i
    return torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=-1)


def _normalize(v, eps):
    return v / (torch.linalg.norm(v, dim=-1, keepdim=True) + eps)


def canonicalize(a, b, p, sp, eps):
    a, b, p, sp = _normalize(a, eps), _normalize(b, eps), _normalize(p, eps), _normalize(sp, eps)
    c = (b * p).sum(-1)[None, :, None]                              # [1,B,1]
    M = torch.zeros(b.shape[0], 3, 3, dtype=F32)                    # the reference's literal (incomplete) skew matrix
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = b[:, 0], -b[:, 2], b[:, 1]
    M[:, 1, 0], M[:, 1, 2], M[:, 2, 0] = b[:, 2], -b[:, 0], -b[:, 1]
    v = (M @ p)[None]                                               # [1,B,3]
    A = a[:, None, :]
    out = v * (A * v).sum(-1, keepdim=True) / (1 + c) + c * A + (A * b[None]).sum(-1, keepdim=True) * p - (A * p).sum(-1, keepdim=True) * b[None]
    mirrored = 2 * (A * sp).sum(-1, keepdim=True) * sp - A
    out = torch.where((1 + c) < eps, mirrored.expand_as(out), out)
    return out / torch.linalg.norm(out, dim=-1, keepdim=True)


def scores(grid, canon, sigma, eps):
    cos = (grid[None, None] * canon[:, :, None, :]).sum(-1)         # f64 [H,O,N]
    geo = torch.acos(torch.clip(cos, -1 + eps, 1 - eps))
    return 1 / torch.exp(geo ** 2 / sigma ** 2)


class ComATorch:
    def __init__(self, H, O, N, size, thres, sigma=0.1, eps=1e-8, p=(0, 0, 1), sp=(0, 1, 0)):
        self.size, self.thres, self.sigma, self.eps = size, thres, sigma, eps
        self.grid = fibonacci_sphere(N)
        self.p, self.sp = torch.tensor(p, dtype=F32), torch.tensor(sp, dtype=F32)
        self.P_h_wrt_o, self.P_o_wrt_h = torch.zeros(H, O, N, dtype=F32), torch.zeros(H, O, N, dtype=F32)
        self.nom, self.den, self.cnt = torch.zeros(H, O, dtype=F32), torch.zeros(H, O, dtype=F32), torch.zeros(H, O, dtype=F32)

    @torch.no_grad()
    def aggregate_sample(self, human_verts, human_normals, obj_verts, obj_normals):
        hv, hn, ov, on = (torch.as_tensor(x).to(F32) for x in (human_verts, human_normals, obj_verts, obj_normals))
        d = torch.sqrt(torch.square(hv[:, None] - ov[None]).sum(-1))
        self.cnt += (d < self.thres).long()
        self.nom += torch.exp(-d / self.size)
        self.den += 1.0
        self.P_h_wrt_o += scores(self.grid, canonicalize(hn, on, self.p, self.sp, self.eps), self.sigma, self.eps)
        self.P_o_wrt_h += scores(self.grid, canonicalize(on, hn, self.p, self.sp, self.eps).permute(1, 0, 2), self.sigma, self.eps)
