#!/usr/bin/env python3
"""HBM traffic of the conv_gemm launches of ONE eager UNet forward from rocprofv3 counter CSVs (FETCH_SIZE / WRITE_SIZE
passes of scripts/pmc_cmd.sh over `scripts/time_unet.py 16 N --eager`): prints (2*FETCH + WRITE) bytes summed over the
kernel family, per forward and per launch (FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import re, sys
fetch, write, n = {}, {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(void sd::conv_gemm_kernel<.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+mean=(\S+)\s+n=(\d+)", line)   # (names may be cut off)
    if m:
        (fetch if m.group(2) == "FETCH_SIZE" else write)[m.group(1)] = float(m.group(3))
        n[m.group(1)] = int(m.group(4))
forwards = int(sys.argv[2])
tot = sum(n[k] * (2 * fetch[k] + write[k]) for k in fetch) * 1024
launches = sum(n.values())
print(f"conv_gemm launches {launches} over {forwards} forwards = {launches / forwards:.0f} per forward")
print(f"(2*FETCH + WRITE) = {tot / forwards / 1e9:.3f} GB per forward, {tot / launches / 1e6:.2f} MB per launch")
for k in sorted(fetch):
    print(f"  {k:55s} n={n[k]:5d}  fetch {fetch[k] * 1024 / 1e6:9.2f} MB (x2)  write {write[k] * 1024 / 1e6:9.2f} MB")
