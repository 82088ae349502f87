"""Python binding of the library-owned models (include/sd_hip.h, coma_amd/csrc/sd_plan.hip): the launch list of a network and its
hipGraph live in libcoma_hip.so; Python decides which layer follows which (unet.py / vae.py record a plan once) and then only asks
for replays.  A saved model (``save``) is everything a caller without Python needs: ``sd_model_load`` + ``sd_unet_forward``."""
from __future__ import annotations

import ctypes as C
import threading

import torch

from .. import _lib

BUF_PERSISTENT, BUF_ZEROED, BUF_IF_NEW = 1, 2, 4


class _Recording(threading.local):
    model = None           # the SdModel THIS THREAD is recording into (ops._p registers every tensor it is handed); thread-local like
                           # the C side's t_plan, so that a second thread's launches are neither registered nor swallowed


_rec = _Recording()


def recording():
    return _rec.model


class SdModel:
    def __init__(self, device, handle=None):
        self.device = torch.device(device)
        self._keep = []                     # tensors the plans point into (a recorded model borrows the caller's buffers)
        if handle is None:
            h = C.c_void_p()
            _lib.check(_lib.lib().sd_model_create(C.byref(h)), "sd_model_create")
            handle = h
        self.h = handle

    @classmethod
    def load(cls, path, device="cuda"):
        h = C.c_void_p()
        with torch.cuda.device(torch.device(device)):
            _lib.check(_lib.lib().sd_model_load(str(path).encode(), C.byref(h)), "sd_model_load")
        return cls(device, h)

    def close(self):
        if self.h is not None:
            _lib.lib().sd_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass

    # ---- registry / bindings
    def register(self, t: torch.Tensor, flags=BUF_PERSISTENT):
        """Register the whole storage behind `t` (views of one storage share a registry entry)."""
        st = t.untyped_storage()
        if st.nbytes() == 0:
            return
        _lib.check(_lib.lib().sd_model_register_buffer(self.h, C.c_void_p(st.data_ptr()), st.nbytes(), flags), "sd_model_register_buffer")
        self._keep.append(t)

    def bind(self, name, t: torch.Tensor):
        assert t.is_contiguous()
        _lib.check(_lib.lib().sd_model_bind(self.h, name.encode(), C.c_void_p(t.data_ptr()), t.numel() * t.element_size()), "sd_model_bind")

    def binding(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().sd_model_binding(self.h, name.encode(), C.byref(p), C.byref(n)), "sd_model_binding")
        return p.value, n.value

    # ---- plans
    def record(self, plan, fn):
        """Run `fn` (Python code that calls the sd_* wrappers of ops.py) with every launch recorded into `plan` instead of issued."""
        _lib.check(_lib.lib().sd_model_record_begin(self.h, plan.encode()), "sd_model_record_begin")
        _rec.model = self
        try:
            fn()
        finally:
            _rec.model = None
            _lib.check(_lib.lib().sd_model_record_end(self.h), "sd_model_record_end")

    def num_launches(self, plan):
        return _lib.lib().sd_model_num_launches(self.h, plan.encode())

    def run(self, plan):
        _lib.check(_lib.lib().sd_model_run(self.h, plan.encode(), _lib.stream_ptr(self.device)), "sd_model_run")

    def prepare(self, plan):
        """Capture + instantiate the plan's hipGraph now (nothing executes)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sd_model_prepare(self.h, plan.encode()), "sd_model_prepare")

    def replay(self, plan):
        _lib.check(_lib.lib().sd_model_replay(self.h, plan.encode(), _lib.stream_ptr(self.device)), "sd_model_replay")

    def save(self, path):
        torch.cuda.synchronize(self.device)
        _lib.check(_lib.lib().sd_model_save(self.h, str(path).encode()), "sd_model_save")

    # ---- network-level entry points (what a C caller uses; device pointers or None = in place)
    def _ptr(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def unet_set_context(self, ctx=None):
        _lib.check(_lib.lib().sd_unet_set_context(self.h, self._ptr(ctx), _lib.stream_ptr(self.device)), "sd_unet_set_context")

    def unet_forward(self, x_in=None, timesteps=None, eps_out=None):
        _lib.check(_lib.lib().sd_unet_forward(self.h, self._ptr(x_in), self._ptr(timesteps), self._ptr(eps_out), _lib.stream_ptr(self.device)),
                   "sd_unet_forward")

    def vae_decode(self, z=None, image_out=None):
        _lib.check(_lib.lib().sd_vae_decode(self.h, self._ptr(z), self._ptr(image_out), _lib.stream_ptr(self.device)), "sd_vae_decode")

    def vae_encode(self, image=None, moments_out=None):
        _lib.check(_lib.lib().sd_vae_encode(self.h, self._ptr(image), self._ptr(moments_out), _lib.stream_ptr(self.device)), "sd_vae_encode")
