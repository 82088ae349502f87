"""CPU: host-side logic of the ComA path (no kernels): dtype rules, state schema, cache bookkeeping,
sharding arithmetic, the 2-rank all-reduce over gloo, and the 'no CPU fallback' guarantee."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from coma_amd import dist as cdist
from coma_amd._lib import ComaHipError
from coma_amd.misc import to_np_torch_recursive
from tests.synth import make_samples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coma(H=6, O=4, N=16):
    from utils.coma import ComA
    return ComA(H, O, N, 0, proximity_settings=dict(spatial_grid_size=0.07, spatial_grid_thres=0.03),
                normal_gaussian_sigma=0.25, eps=1e-10, device="cpu")


def test_dtype_table_matches_reference(golden):
    table = []
    for dt in [np.float64, np.float32, np.float16, np.int64, np.int32, np.int16, np.uint8, np.bool_]:
        t = to_np_torch_recursive(np.zeros(2, dt), use_torch=True, device="cpu")
        n = to_np_torch_recursive(torch.zeros(2, dtype=t.dtype), use_torch=False, device="cpu")
        table.append(f"{np.dtype(dt).name}->{t.dtype}->{n.dtype}")
    assert table == list(golden["g12_dtype_table"])


def test_recursive_walk_on_nested_containers():
    x = dict(a=np.zeros(2, np.float64), b=[np.zeros(1, np.int32), dict(c=torch.zeros(1, dtype=torch.float64))], d="s", e=3)
    y = to_np_torch_recursive(x, use_torch=True, device="cpu")
    assert y["a"].dtype == torch.float32 and y["b"][0].dtype == torch.int64 and y["b"][1]["c"].dtype == torch.float32
    assert y["d"] == "s" and y["e"] == 3


def test_state_schema_matches_reference_export(golden):
    c = _coma(32, 8, 250)
    exp = c.export()
    assert sorted(exp.keys()) == list(golden["g4_export_keys"])
    got = [f"{k}:{getattr(exp[k], 'dtype', type(exp[k]).__name__)}" for k in sorted(exp.keys())]
    assert got == list(golden["g4_export_dtypes"])
    assert c.canon_normal_grid.dtype == torch.float64          # f64 while learning
    assert np.array_equal(c.canon_normal_grid.numpy(), golden["g1_sphere250"])


def test_export_pickle_is_loadable_and_names_utils_coma(tmp_path):
    c = _coma()
    p = tmp_path / "c.pickle"
    c.export(str(p))
    raw = open(p, "rb").read()
    assert b"utils.coma" in raw and b"negative_exp" in raw     # the reference un-pickles this symbol path
    d = pickle.load(open(p, "rb"))
    assert d["contact_dist_func"].func.__name__ == "negative_exp"
    c2 = _coma()
    c2.load(str(p))
    assert c2.canon_normal_grid.dtype == torch.float32          # f32 after load, as in the reference


def test_reference_written_pickle_loads_on_host():
    c = _coma(6, 4, 16)
    c.load(os.path.join(ROOT, "tests", "golden", "ref_coma_small.pickle"))
    assert c.used_count == 2 and tuple(c.prob_grid_canon_human_wrt_obj.shape) == (6, 4, 16)
    assert float(c.contact_dist_expectation_grid_denom.min()) == 2.0


def test_cache_bookkeeping_and_shape_asserts():
    c = _coma()
    smp = make_samples(6, 4, 2, 0, 0.03)
    for s in smp:
        c.register_sample_to_cache(**s)
    assert c.cache_count == 2 and list(c.cache) == ["00000", "00001"]
    bad = dict(smp[0], human_verts=smp[0]["human_verts"][:5])
    with pytest.raises(AssertionError):
        c.aggregate_single_sample(**bad)
    with pytest.raises(AssertionError):
        c.compute_contact_map("nope")


def test_no_cpu_fallback():
    c = _coma()
    with pytest.raises(ComaHipError):
        c.aggregate_single_sample(**make_samples(6, 4, 1, 0, 0.03)[0])
    with pytest.raises(ComaHipError):
        c.compute_contact_map("human")
    from utils.coma_occupancy import ComA_Occupancy
    o = ComA_Occupancy(scale_tolerance=3.0, human_res=4, obj_res=1, normal_res=0, spatial_res=4, device="cpu")
    with pytest.raises(ComaHipError):
        o.return_aggregated_spatial_grids()


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "coma_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_voxelgrid_host_matches_golden(golden):
    from utils.coma_occupancy import load_voxelgrid
    for R in (4, 30):
        g, ig, md = load_voxelgrid(2.4, R, [0, 0, 0])
        assert g.dtype == np.float64 and md["voxel_size"] == float(golden[f"g8_voxel_{R}"])
        assert np.array_equal(np.stack([g[0, :, 0, 0], g[1, 0, :, 0], g[2, 0, 0, :]]), golden[f"g8_axis_{R}"])


def test_shard_arithmetic():
    assert [cdist.shard_slice(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sum(b - a for a, b in (cdist.shard_slice(512, r, 8) for r in range(8))) == 512
    # the reference's slice rule: 512 items on 8 ranks -> 65 x 7 + 57 (src/generation/inpaint.py:271-274)
    sizes = [b - a for a, b in (cdist.reference_slice(512, r, 8) for r in range(8))]
    assert sizes == [65] * 7 + [57]


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from coma_amd import dist as cdist
from utils.coma import ComA
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
c = ComA(5, 3, 8, 0, proximity_settings=dict(spatial_grid_size=0.07, spatial_grid_thres=0.03), device="cpu")
g = torch.Generator().manual_seed(rank)
for k in c._STATE_KEYS:
    getattr(c, k).copy_(torch.rand(getattr(c, k).shape, generator=g))
c.used_count = 3 + rank
c.all_reduce()
exp = {}
for k in c._STATE_KEYS:
    tot = 0
    for r in range(world):
        gg = torch.Generator().manual_seed(r)
        for kk in c._STATE_KEYS:
            t = torch.rand(getattr(c, kk).shape, generator=gg)
            if kk == k:
                tot = tot + t
    assert torch.allclose(getattr(c, k), tot), k
assert c.used_count == sum(3 + r for r in range(world))
t = torch.tensor([1.0, float("nan") if rank == 1 else 0.5, -2.0 + rank])
cdist.all_reduce_max_nan(t)
assert t[0] == 1.0 and torch.isnan(t[1]) and t[2] == -2.0 + (world - 1)
dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_two_rank_all_reduce_over_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, out


_OCC_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from coma_amd import dist as cdist
from oracle import coma_oracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H, R = 11, 6                                   # 11 rows over 2 ranks: 6 + 5 (ragged shards)
oracle = orc.OccupancyOracle(H, R, 3.0)
rng = np.random.default_rng(3)
for _ in range(5):
    hv = rng.uniform(-1.0, 1.0, size=(H, 3))
    hv[4] = [9.0, 9.0, 9.0]                     # vertex 4 never enters the grid -> its row is 0/0 = NaN after normalising
    oracle.aggregate_sample(hv, np.zeros((1, 3)))
lo, hi = cdist.shard_slice(H, rank, world)


class Shard:                                    # the device object's contract, on CPU tensors (the kernels need a GPU)
    def __init__(self):
        self.spatial_occupancy_grids = torch.from_numpy(oracle.occ[lo:hi].copy())
        self.spatial_grid = torch.zeros(3, R, R, R)
    def normalize_prob_grid_for_spatials(self):
        g = self.spatial_occupancy_grids
        g /= g.reshape(g.shape[0], -1).sum(-1)[:, None, None, None]
    def return_aggregated_spatial_grids(self, human_indices=None):
        self.normalize_prob_grid_for_spatials()
        g = self.spatial_occupancy_grids if human_indices is None else self.spatial_occupancy_grids[list(human_indices)]
        return torch.max(g, dim=0).values
    def reduce_keep_raw(self, human_indices=None, want_raw=True):
        raw = self.spatial_occupancy_grids.clone() if want_raw else None
        if human_indices is not None and len(human_indices) == 0:
            self.normalize_prob_grid_for_spatials()
            return raw, torch.full((R, R, R), float("-inf"))
        return raw, self.return_aggregated_spatial_grids(human_indices)


for sel in (None, [0, 1, 2, 9], [6, 7]):        # all rows (NaN row poisons the field), rows from both shards, rows of one shard only
    full, field = cdist.occupancy_rows_reduce(Shard(), H, human_indices=sel)
    ref_counts = oracle.occ.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = ref_counts / ref_counts.reshape(H, -1).sum(-1)[:, None, None, None]
    ref = np.max(norm if sel is None else norm[sel], axis=0)     # np.max propagates NaN like torch.max
    assert np.array_equal(field.numpy(), ref, equal_nan=True), sel
    assert (sel is None) == bool(np.isnan(ref).any())
    if rank == 0:
        assert np.array_equal(full.numpy(), oracle.occ)             # raw counts of every row, in order
    else:
        assert full is None
dist.destroy_process_group()
print("RANK_OK", rank)
"""


def test_row_sharded_occupancy_reduce_over_gloo(tmp_path):
    """SURVEY.md 8e-3 as src/coma/extract_coma.py runs it with WORLD_SIZE > 1: rows sharded, raw rows gathered to rank 0 for the
    export, local normalise + max, NaN-propagating MAX all-reduce of the [R,R,R] field."""
    script = tmp_path / "wocc.py"
    script.write_text(_OCC_WORKER)
    port = 31500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, out


_OCC_SHARD_EXPORT_WORKER = r"""
import os, pickle, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from coma_amd import dist as cdist
from coma_amd.coma_occupancy import ComA_Occupancy
from oracle import coma_oracle as orc
out_dir = sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H, R = 11, 6                                   # 11 rows over 2 ranks: 6 + 5 (ragged shards)
oracle = orc.OccupancyOracle(H, R, 3.0)
rng = np.random.default_rng(3)
for _ in range(5):
    oracle.aggregate_sample(rng.uniform(-1.0, 1.0, size=(H, 3)), np.zeros((1, 3)))
lo, hi = cdist.shard_slice(H, rank, world)
kw = dict(scale_tolerance=3.0, obj_res=1, normal_res=0, spatial_res=R, device="cpu")      # host-side state only: no kernel runs
save_pth = os.path.join(out_dir, "key:total.pickle")

shard = ComA_Occupancy(human_res=hi - lo, **kw)
shard.spatial_occupancy_grids = torch.from_numpy(oracle.occ[lo:hi].copy())
shard.used_count = 5
shard.export(save_pth=save_pth, shard=(rank, world, H))
assert os.path.exists(os.path.join(out_dir, f"key:total_rank{rank}.pickle")) and not os.path.exists(save_pth)
dist.barrier()

files = ComA_Occupancy.shard_files(save_pth)
assert [os.path.basename(f) for f in files] == ["key:total_rank0.pickle", "key:total_rank1.pickle"]
single = ComA_Occupancy(human_res=H, **kw)
single.spatial_occupancy_grids = torch.from_numpy(oracle.occ.copy())
single.used_count = 5
ref = single.export()                           # what a single process writes (utils/coma_occupancy.py:315-330)
got = ComA_Occupancy.assemble_shards(files)
assert list(got.keys()) == list(ref.keys()), (list(got.keys()), list(ref.keys()))
for k in ref:
    a, b = ref[k], got[k]
    if isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k
    elif isinstance(a, dict):
        assert pickle.dumps(a) == pickle.dumps(b), k
    else:
        assert a == b, k
with open(files[rank], "rb") as fh:
    mine = pickle.load(fh)
assert mine["row_shard"] == (rank, world, lo, hi) and mine["human_res"] == H and set(mine) == set(ref) | {"row_shard"}
assert np.array_equal(mine["spatial_occupancy_grids"], oracle.occ[lo:hi])

# loader: own shard only (same world), the whole grid from the shard set, and a row slice of a single file
o = ComA_Occupancy(human_res=hi - lo, **kw)
o.load(save_pth, shard=(rank, world))
assert o.human_res == hi - lo and np.array_equal(o.spatial_occupancy_grids.numpy(), oracle.occ[lo:hi]) and not hasattr(o, "row_shard")
o = ComA_Occupancy(human_res=H, **kw)
o.load(save_pth)
assert o.human_res == H and np.array_equal(o.spatial_occupancy_grids.numpy(), oracle.occ) and o.used_count == 5
o3 = ComA_Occupancy(human_res=1, **kw)
o3.load(save_pth, shard=(rank, 3))               # another world size: assembled, then sliced
l3, h3 = cdist.shard_slice(H, rank, 3)
assert np.array_equal(o3.spatial_occupancy_grids.numpy(), oracle.occ[l3:h3])
dist.barrier()
if rank == 0:
    single.export(save_pth=save_pth)
dist.barrier()
o = ComA_Occupancy(human_res=hi - lo, **kw)
o.load(save_pth, shard=(rank, world))            # the single file wins when it exists
assert o.human_res == hi - lo and np.array_equal(o.spatial_occupancy_grids.numpy(), oracle.occ[lo:hi])
try:
    ComA_Occupancy(human_res=H, **kw).load(os.path.join(out_dir, "missing.pickle"))
    raise SystemExit("expected FileNotFoundError")
except FileNotFoundError:
    pass
dist.destroy_process_group()
print("RANK_OK", rank)
"""


def test_row_sharded_occupancy_export_over_gloo(tmp_path):
    """SURVEY.md 7 "Memory at config 5": every rank pickles its own row slice (`..._rank{r}.pickle`, same keys as the reference's
    export, utils/coma_occupancy.py:315-330); the shards re-assemble to the single-process export bit for bit and the loader
    takes either form."""
    script = tmp_path / "wshard.py"
    script.write_text(_OCC_SHARD_EXPORT_WORKER)
    port = 33500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, out


def test_presets_match_reference_tables():
    import json
    from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT as Q
    from constants.coma.quant import QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT as N
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "presets.json")))
    norm = lambda d: json.loads(json.dumps(d))            # tuples -> lists, as stored
    for k, v in ref["qual"].items():
        assert norm(Q[k]) == v, k
    assert norm(N) == ref["quant"]
    # aliases for the keys scripts/learn_coma.sh passes
    assert Q["qual:backpack_human"] == Q["qual:backpack_human_contact"]


def test_inference_cli_flags_match_reference():
    """Same argparse surface as the reference's src/coma/inference.py:150-161."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "src", "coma", "inference.py"), "--help"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--supercategory", "--category", "--coma_path", "--visualize_type", "--smplx_downsample_pth",
                 "--asset_downsample_pth", "--hyperparams_key", "--output_dir", "--seed"):
        assert flag in out.stdout


def test_bench_starts_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a launcher re-executes under torch.distributed.run with one rank per GPU on 127.0.0.1
    (the driver starts N = 1 as a plain python call and must be able to start N > 1 the same way)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["COMA_BENCH_PRINT_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    # under a launcher with a different world size the script refuses instead of asserting
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
