"""Build libcoma_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m coma_amd.build            # incremental
    python -m coma_amd.build --force

One object per .hip file under coma_amd/csrc (compiled in parallel), linked into
coma_amd/libcoma_hip.so.  -ffp-contract=off keeps the bit-exact sections (distance tests, f64 argmin)
free of compiler-introduced FMAs; the hot loops ask for FMAs explicitly with fmaf().
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcoma_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA results land directly in VGPRs (gfx950 has one unified register file); without it hipcc
# parks accumulators in AGPRs and pays a v_accvgpr_read/write per element every time VALU code (softmax, epilogues)
# touches them -- 159 extra instructions per attention key tile.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-variable", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-Rpass-analysis=kernel-resource-usage"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "coma_hip.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sd_hip.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "seg_hip.h"))
    jobs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        # no kernel of this library may touch scratch: a spilled accumulator tile silently costs far more than it saves, and a
        # 2-VGPR spill in the 256 x 320 epilogue was observed to come back corrupted on gfx950 (round 2) -- refuse to build it
        name = ""
        for line in r.stderr.splitlines():
            if "Function Name:" in line:
                name = line.split("Function Name:")[1].split()[0]
            elif "ScratchSize [bytes/lane]:" in line and int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) > 0:
                os.remove(cmd[cmd.index("-o") + 1])
                raise RuntimeError(f"kernel {name} spills to scratch ({line.strip()}); reduce its register pressure")
        diag = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l)
        if verbose and diag.strip():
            print(diag, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in _sources()]
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
