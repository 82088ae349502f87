#!/usr/bin/env python3
"""Phase stamps of sd_conv3x3_halo_f16 (tuning aid; needs the -DHALO_DBG build: scripts/build_variant.sh halo_dbg -DHALO_DBG, run with
COMA_HIP_LIB=coma_amd/_ab/halo_dbg.so).  Per workgroup (wave 0): cycles entry -> first commit -> end of K loop -> end of epilogue, cycles
parked in the per-slice wait + barrier, cycles in the commits."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coma_amd import _lib
from coma_amd.sd import ops
dev = "cuda:0"
B, H, W, n = 8, 512, 512, 128
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = B * H * W
x = torch.randn(M, Cc, device=dev).half()
w = (torch.randn(n, 9 * Cc, device=dev) * (9 * Cc) ** -0.5).half()
bias, res = torch.randn(n, device=dev).half(), torch.randn(M, n, device=dev).half()
table = torch.rand(B * Cc * 2, device=dev) + 0.5
out = torch.empty(M, n, dtype=torch.float16, device=dev)
cs = torch.zeros(M // 256, 2, n, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.conv3x3_halo(x, w, out, batch=B, h=H, w_=W, c=Cc, bias=bias, res=res, gn_affine=table, silu=True, colstats=cs)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.conv3x3_halo(x, w, out, batch=B, h=H, w_=W, c=Cc, bias=bias, res=res, gn_affine=table, silu=True, colstats=cs)
e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) * 1e3
nb = 8192
st = np.zeros((nb, 8), dtype=np.uint64)
lib = _lib.lib()
lib.sd_halo_debug.restype = C.c_int
assert lib.sd_halo_debug(st.ctypes.data_as(C.c_void_p), nb) == 0
st = st.astype(np.int64)
span = int(st[:, 3].max() - st[:, 0].min())
tot = st[:, 3] - st[:, 0]
print(f"C={Cc}: {us:.1f} us, span {span / 1e3:.0f} kcyc -> {span / us / 1e3:.2f} GHz; per workgroup (median, kcyc): total {np.median(tot) / 1e3:.1f}  "
      f"entry->first commit {np.median(st[:, 1] - st[:, 0]) / 1e3:.1f}  K loop {np.median(st[:, 2] - st[:, 1]) / 1e3:.1f}  "
      f"epilogue {np.median(st[:, 3] - st[:, 2]) / 1e3:.1f} | parked in slice wait+barrier {np.median(st[:, 4]) / 1e3:.1f}  commits {np.median(st[:, 5]) / 1e3:.1f}")
# occupancy of the two slots per CU: sum of workgroup lifetimes / (512 slots x kernel span)
print(f"  slot occupancy: sum of workgroup lifetimes / (2 x 256 CUs x span) = {tot.sum() / (512.0 * span):.2f}; first entry -> last exit {span / 1e3:.0f} kcyc")
print(f"  MFMA issue floor per workgroup-wave: {9 * (Cc // 64) * 64 * 16 / 1e3:.1f} kcyc; workgroups resident per CU: 2; rounds {nb / 512:.0f}")
