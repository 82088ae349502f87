"""CPU: host rules of the learning CLI (src/coma/extract_coma.py) -- discovery, mainprompt rule, post-filter
membership, sentinel skipping, index-range parsing -- on a synthetic results/ tree."""
import json
import os
import pickle

import numpy as np
import pytest

from src.coma import extract_coma as ec


def _tree(tmp_path):
    root = tmp_path / "human_sample"
    def put(sc, c, asset, view, mask, prompt, i, payload):
        d = root / sc / c / asset / view / mask / prompt
        d.mkdir(parents=True, exist_ok=True)
        pickle.dump(payload, open(d / f"{i:06}.pickle", "wb"))
    ok = dict(verts=np.zeros((4, 3)), faces=np.zeros((1, 3), np.int64))
    put("BEHAVE", "backpack", "behave_asset", "view:00000", "00001", "1 person wears the backpack", 0, ok)
    put("BEHAVE", "backpack", "behave_asset", "view:00000", "00001", "1 person wears the backpack, full body", 1, ok)
    put("BEHAVE", "backpack", "behave_asset", "view:00001", "00002", "1 person wears the backpack", 2, "TOO LITTLE INLIERS")
    put("BEHAVE", "backpack", "behave_asset", "view:00001", "00002", "total:1 person wears the backpack", 3, ok)
    put("3D:FUTURE", "Lounge Chair : Cafe Chair", "chair0", "view:00000", "00000", "1 person sits on the chair", 0, ok)
    return str(root)


def test_mainprompt_rule_and_discovery(tmp_path):
    root = _tree(tmp_path)
    assert ec.mainprompt_of("1 person wears the backpack, full body") == "1 person wears the backpack"
    assert ec.mainprompt_of("total:1 person wears the backpack") == "total"
    scams = ec.discover_scams(root)
    assert scams == [("3D/FUTURE", "Lounge Chair / Cafe Chair", "chair0", "1 person sits on the chair"),
                     ("BEHAVE", "backpack", "behave_asset", "1 person wears the backpack"), ("BEHAVE", "backpack", "behave_asset", "total")]
    assert ec.discover_scams(root, categories=["backpack"], prompts=["total"]) == [("BEHAVE", "backpack", "behave_asset", "total")]


def test_postfilter_and_sentinels(tmp_path):
    root = _tree(tmp_path)
    scam = ("BEHAVE", "backpack", "behave_asset", "1 person wears the backpack")
    # without post-filter: the sentinel string is skipped silently
    kept = ec.collect_inputs(scam, root, ec.PostFilter(False, None), enable_postfilter=False)
    assert sorted(os.path.basename(p) for p in kept) == ["000000.pickle", "000001.pickle"]   # path order: "," < "/"
    # with post-filter: only listed (view, mask, prompt, id) tuples survive; a listed sentinel is an error upstream
    pf_dir = tmp_path / "pf" / "BEHAVE" / "backpack" / "behave_asset"
    pf_dir.mkdir(parents=True)
    json.dump([["view:00000", "00001", "1 person wears the backpack, full body", "000001"]],
              open(pf_dir / "1 person wears the backpack.json", "w"))
    kept = ec.collect_inputs(scam, root, ec.PostFilter(True, str(tmp_path / "pf")), enable_postfilter=True)
    assert [os.path.basename(p) for p in kept] == ["000001.pickle"]
    with pytest.raises(AssertionError):
        ec.collect_inputs(("BEHAVE", "backpack", "behave_asset", "total"), root, ec.PostFilter(True, str(tmp_path / "pf")), True)


def test_selected_object_indices_and_flags():
    assert ec.parse_selected_object_indices("") is None
    assert ec.parse_selected_object_indices("21 22") == [21, 22]
    assert ec.parse_selected_object_indices("3 21-25 22") == [3, 21, 22, 23, 24, 25]
    flags = {a.option_strings[0] for a in ec.build_parser()._actions if a.option_strings}
    for f in ("--supercategories", "--categories", "--prompts", "--camera_dir", "--human_params_dir", "--asset_downsample_dir",
              "--human_postfilter_dir", "--human_sample_dir", "--coma_save_dir", "--affordance_save_dir", "--smplx_canon_obj_pth",
              "--hyperparams_key", "--visualize", "--vis_example_num", "--interactive", "--vis_interactive", "--fovy",
              "--tmp_cache_dir", "--selected_object_indices", "--scale_tolerance", "--skip_done", "--seed"):
        assert f in flags, f
    assert "qual:backpack_human_contact" in ec.build_parser()._option_string_actions["--hyperparams_key"].choices


def test_vertex_face_csr_is_ascending():
    from coma_amd.ingest import vertex_face_csr
    faces = np.array([[0, 1, 2], [2, 1, 3], [0, 2, 3]])
    off, vf = vertex_face_csr(faces, 5)
    assert list(off) == [0, 2, 4, 7, 9, 9]
    assert list(vf[off[2]:off[3]]) == [0, 1, 2] and list(vf[off[0]:off[1]]) == [0, 2]
