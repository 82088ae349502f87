"""Host helpers that fix the dtype rules of the ComA path.

Mirrors (same names, argument meaning and results):
  to_np_torch_recursive   reference utils/misc.py:14-63   floats -> f32, ints -> i64, both directions
  get_3d_indexgrid_ijk    reference utils/misc.py:66-83
  normalize_vectors_np    reference utils/transformations.py:8-11
  seed_everything         reference utils/reproducibility.py:11-20
Pinned by tests/golden G12 (dtype table).
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

try:  # easydict is optional in this image; the reference treats EasyDict exactly like dict
    from easydict import EasyDict  # type: ignore
except Exception:  # pragma: no cover
    class EasyDict(dict):
        """attribute-style dict; pickles under the name `easydict.EasyDict` so that files written here open with the real package"""
        __module__ = "easydict"

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k) from None

        def __setattr__(self, k, v):
            self[k] = v

    import sys as _sys
    import types as _types
    if "easydict" not in _sys.modules:           # lets pickle resolve easydict.EasyDict when reading such files back here
        _m = _types.ModuleType("easydict")
        _m.EasyDict = EasyDict
        _sys.modules["easydict"] = _m

_CONTAINERS = (dict, EasyDict, np.ndarray, torch.Tensor, list)
_T_FLOAT = (torch.float64, torch.float32, torch.float16, torch.bfloat16)
_T_INT = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64)
_N_FLOAT = (np.float32, np.float16, np.float64)
_N_INT = (np.int64, np.int32, np.int16)


def to_np_torch_recursive(X, use_torch=True, device="cuda", np_float_type=np.float32, np_int_type=np.int64,
                          torch_float_type=torch.float32, torch_int_type=torch.int64):
    """Walk dicts/lists; arrays <-> tensors; every float becomes f32 and every int i64.

    Note (as in the reference): nested calls use the *default* dtypes, and uint8/int8 ndarrays and
    bool arrays are left alone on the numpy side.
    """
    t = type(X)
    if t in (dict, EasyDict):
        for k in X.keys():
            if type(X[k]) in _CONTAINERS:
                X[k] = to_np_torch_recursive(X[k], use_torch, device)
    elif t is list:
        for i in range(len(X)):
            if type(X[i]) in _CONTAINERS:
                X[i] = to_np_torch_recursive(X[i], use_torch, device)
    elif t is np.ndarray:
        if use_torch:
            X = torch.tensor(X, device=device)
    elif t is torch.Tensor:
        X = X.to(device) if use_torch else X.detach().cpu().numpy()

    if type(X) is torch.Tensor:
        if X.dtype in _T_FLOAT:
            X = X.type(torch_float_type)
        elif X.dtype in _T_INT:
            X = X.type(torch_int_type)
    elif type(X) is np.ndarray:
        if X.dtype in _N_FLOAT:
            X = X.astype(np_float_type)
        elif X.dtype in _N_INT:
            X = X.astype(np_int_type)
    return X


def get_3d_indexgrid_ijk(N_x, N_y, N_z, raveled=False):
    idx = np.mgrid[0:N_x, 0:N_y, 0:N_z]
    if raveled:
        idx = np.stack([idx[0].ravel() + 1, idx[1].ravel() + 1, idx[2].ravel() + 1], axis=-1)
    return idx


def normalize_vectors_np(vecs, eps=1e-8):
    assert vecs.ndim == 2 and vecs.shape[-1] == 3
    return vecs / (np.sqrt(np.sum(np.square(vecs), axis=-1, keepdims=True)) + eps)


def normalize_vectors_torch(vecs, eps=1e-8):
    assert vecs.ndim == 2 and vecs.shape[-1] == 3
    return vecs / (torch.sqrt(torch.sum(torch.square(vecs), dim=-1, keepdim=True)) + eps)


def seed_everything(seed: int, workers: bool = False):
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PL_SEED_WORKERS"] = f"{int(workers)}"
    return seed
