"""GPU: the library-owned models (include/sd_hip.h "Models", coma_amd/csrc/sd_plan.hip).
(1) the recorded plan replayed by the library (eager list and hipGraph) gives the same bits as the Python launch list;
(2) a model file written by sd_model_save is loaded and run by a C program that never touches Python -- sd_model_load +
    sd_unet_set_context / sd_unet_forward, sd_vae_decode, sd_vae_encode -- and returns the same bits as the in-process network
    (these three are C entry points now)."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY_UNET = dict(in_channels=9, out_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2, heads=8, cross_attention_dim=768,
                 groups=32, down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True))
TINY_VAE = dict(latent_channels=4, block_out_channels=(64, 128, 128, 128), layers_per_block=2, groups=32, scaling_factor=0.18215)


@pytest.fixture(scope="module")
def c_runner(tmp_path_factory, hip_lib):
    exe = tmp_path_factory.mktemp("c") / "run_model"
    libdir = os.path.join(ROOT, "coma_amd")
    # plain gcc: the caller is C, it needs only the two headers, libcoma_hip.so and the HIP runtime for its own buffers
    cmd = ["gcc", os.path.join(ROOT, "tests", "c", "run_model.c"), "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", "-L" + libdir, "-lcoma_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return str(exe)


def _unet(batch=2, hw=16, **kw):
    from coma_amd.sd import weights
    from coma_amd.sd.unet import HipUNet2DConditionModel
    st = weights.random_state(weights.unet_shapes(TINY_UNET), seed=0)
    return HipUNet2DConditionModel(st, batch=batch, height=hw, width=hw, device=DEV, cfg=TINY_UNET, **kw)


def test_recorded_plan_equals_python_launch_list(hip_lib):
    g = torch.Generator().manual_seed(0)
    ctx, x, t = torch.randn(2, 77, 768, generator=g), torch.randn(2, 9, 16, 16, generator=g), torch.tensor([961.0, 961.0])
    eager = _unet(use_graph=False)
    ref = eager(x, t, encoder_hidden_states=ctx)[0]
    lib = _unet(use_graph=True)
    out1 = lib(x, t, encoder_hidden_states=ctx)[0]                    # first replay: records, warms up, captures
    out2 = lib(x, t, encoder_hidden_states=ctx)[0]                    # graph replay
    assert lib.g.model.num_launches("step") >= len(lib.g.launches) and lib.g.model.num_launches("context") >= len(lib.gc.launches)
    assert torch.equal(ref, out1) and torch.equal(ref, out2)
    lib.g.run_recorded()                                              # the recorded list launched natively, no graph
    torch.cuda.synchronize()
    out3 = torch.empty_like(ref)
    from coma_amd.sd import ops
    ops.nhwc_to_nchw(lib.eps, out3, batch=2, c=4, hw=256, ld=64)
    assert torch.equal(ref, out3)


def test_c_program_runs_saved_unet_and_vae(tmp_path, c_runner, hip_lib):
    from coma_amd.sd import weights
    from coma_amd.sd.vae import HipVaeDecoder, HipVaeEncoder
    g = torch.Generator().manual_seed(1)
    # ---- UNet: batch 2 (one image, CFG), 16 x 16 latents
    unet = _unet(cfg_shared_prefix=True)
    ctx = torch.randn(2, 77, 768, generator=g).half()
    x_in = torch.zeros(2, 256, 64, dtype=torch.float16)
    x_in[0, :, :9] = torch.randn(256, 9, generator=g).half()
    x_in[1] = x_in[0]                                                 # the shared CFG prefix wants identical halves
    t = torch.tensor([961.0, 961.0])
    unet.set_context(ctx.to(DEV))
    unet.x_in.copy_(x_in.to(DEV))
    unet.timesteps.copy_(t.to(DEV))
    ref = unet.forward_static().cpu().numpy().copy()
    unet.save(tmp_path / "unet.sdm")
    for name, arr in (("ctx", ctx), ("x_in", x_in), ("t", t)):
        arr.numpy().tofile(tmp_path / f"{name}.bin")
    r = subprocess.run([c_runner, "unet", str(tmp_path / "unet.sdm"), str(tmp_path / "ctx.bin"), str(tmp_path / "x_in.bin"),
                        str(tmp_path / "t.bin"), str(tmp_path / "eps.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "eps.bin", dtype=np.float16).reshape(ref.shape)
    assert np.array_equal(got, ref) and np.isfinite(got.astype(np.float32)).all() and np.abs(got[:, :4]).max() > 0
    del unet
    # ---- VAE decoder / encoder at 64 x 64 pixels
    vst = weights.random_state(weights.vae_shapes(TINY_VAE), seed=1)
    dec = HipVaeDecoder(vst, 1, latent_h=8, latent_w=8, device=DEV, cfg=TINY_VAE)
    z = torch.zeros(1, 64, 64, dtype=torch.float16)
    z[..., :4] = torch.randn(1, 64, 4, generator=g).half()
    dec.z.copy_(z.to(DEV))
    img = dec.decode_static().cpu().numpy().copy()
    dec.save(tmp_path / "dec.sdm")
    z.numpy().tofile(tmp_path / "z.bin")
    r = subprocess.run([c_runner, "decode", str(tmp_path / "dec.sdm"), str(tmp_path / "z.bin"), str(tmp_path / "img.bin")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(tmp_path / "img.bin", dtype=np.float16).reshape(img.shape), img)
    enc = HipVaeEncoder(vst, 1, height=64, width=64, device=DEV, cfg=TINY_VAE)
    x = torch.zeros(1, 4096, 64, dtype=torch.float16)
    x[..., :3] = (torch.rand(1, 4096, 3, generator=g) * 2 - 1).half()
    enc.x.copy_(x.to(DEV))
    mom = enc.encode_static().cpu().numpy().copy()
    enc.save(tmp_path / "enc.sdm")
    x.numpy().tofile(tmp_path / "x.bin")
    r = subprocess.run([c_runner, "encode", str(tmp_path / "enc.sdm"), str(tmp_path / "x.bin"), str(tmp_path / "mom.bin")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(tmp_path / "mom.bin", dtype=np.float16).reshape(mom.shape), mom)


def test_model_python_roundtrip_and_errors(tmp_path, hip_lib):
    """sd_model_load from Python gives the same network; unknown plans / bindings and unregistered pointers are errors, not crashes."""
    from coma_amd._lib import ComaHipError
    from coma_amd.sd.model import SdModel
    unet = _unet()
    g = torch.Generator().manual_seed(2)
    ctx, x, t = torch.randn(2, 77, 768, generator=g), torch.randn(2, 9, 16, 16, generator=g), torch.tensor([500.0, 500.0])
    ref = unet(x, t, encoder_hidden_states=ctx)[0]
    unet.save(tmp_path / "u.sdm")
    # scratch is not part of a saved model: the file holds the constants in kernel layout -- about the size of the state
    # dict (re-laid-out copies, a few concatenations) -- and neither the 64 MiB split-K workspace nor any activation buffer, although
    # every recorded launch points into them
    state_bytes = sum(v.numel() * 2 for v in unet.s.values())
    size = os.path.getsize(tmp_path / "u.sdm")
    assert 0.8 * state_bytes < size < 1.6 * state_bytes + (8 << 20), (size, state_bytes)
    m = SdModel.load(tmp_path / "u.sdm", DEV)
    assert m.num_launches("step") == unet.g.model.num_launches("step") and m.num_launches("nope") == -1
    eps = torch.empty_like(unet.eps)
    m.unet_set_context(unet.ctx.clone())
    m.unet_forward(unet.x_in.clone(), unet.timesteps.clone(), eps)
    torch.cuda.synchronize()
    assert torch.equal(eps, unet.eps)
    with pytest.raises(ComaHipError):
        m.replay("nope")
    with pytest.raises(ComaHipError):
        m.binding("nope")
    with pytest.raises(ComaHipError):
        SdModel.load(tmp_path / "missing.sdm", DEV)
    bad = SdModel(DEV)
    stray = torch.zeros(64, 320, dtype=torch.float16, device=DEV)
    out = torch.zeros(64, 320, dtype=torch.float16, device=DEV)
    gm, bt = torch.ones(320, dtype=torch.float16, device=DEV), torch.zeros(320, dtype=torch.float16, device=DEV)
    import coma_amd.sd.model as mm
    from coma_amd.sd import ops
    bad.record("p", lambda: ops.layernorm(stray, gm, bt, out, rows=64, c=320))
    assert bad.num_launches("p") == 1
    bad.run("p")                                                      # runs from the caller's buffers
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0                              # LayerNorm of zeros with beta = 0
    assert mm.recording() is None


def test_model_file_is_verified_before_it_is_trusted(tmp_path, hip_lib):
    """Format version 3 (checks of ADVICE r3): a truncated file, a file with one flipped byte in the launch records and one with a flipped byte
    in the weights are all refused by sd_model_load (size / checksum), nothing is launched; and a buffer first registered as scratch,
    later registered SD_BUF_PERSISTENT (constants filled outside a plan), is saved with the model; a range bridging two registered
    buffers leaves one entry."""
    from coma_amd._lib import ComaHipError
    from coma_amd.sd import ops
    from coma_amd.sd.model import BUF_PERSISTENT, SdModel
    m = SdModel(DEV)
    big = torch.zeros(3, 64, 320, dtype=torch.float16, device=DEV)              # one allocation, three row blocks
    x, gamma_beta, out = big[0], big[1], big[2]
    import ctypes
    from coma_amd import _lib

    def reg(t, flags):                                                           # the C entry point on exactly the bytes of `t`
        _lib.check(_lib.lib().sd_model_register_buffer(m.h, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), flags),
                   "sd_model_register_buffer")
    m._keep.append(big)
    reg(x, 0)                                                                    # scratch ...
    reg(out, 0)
    reg(big[0:2, 32:], 0)                                                        # ... a range bridging x and gamma_beta, then all three:
    reg(big, 0)                                                                  # one entry is left (a save would refuse overlapping ones)
    gamma_beta[0].fill_(2.0)                                                     # constants filled OUTSIDE a plan
    gamma_beta[1].fill_(0.5)
    reg(gamma_beta, BUF_PERSISTENT)                                              # covered already: the flag must still be OR-ed in
    x.copy_(torch.randn(64, 320, device=DEV).half())
    m.bind("x", x)
    m.bind("out", out)
    m.record("p", lambda: ops.layernorm(x, gamma_beta[0], gamma_beta[1], out, rows=64, c=320))
    m.run("p")
    torch.cuda.synchronize()
    want = out.clone()
    xin = x.clone()
    path = tmp_path / "m.sdm"
    m.save(path)
    blob = path.read_bytes()
    assert blob[:8] == b"SDMODEL3" and int.from_bytes(blob[8:16], "little") == len(blob)
    # the round trip keeps the constants (gamma = 2, beta = 0.5 live in the persistent range)
    m2 = SdModel.load(path, DEV)
    p, n = m2.binding("x")
    po, no = m2.binding("out")
    assert n == xin.numel() * 2 and no == n
    _lib.check(_lib.lib().sd_copy_d2d(ctypes.c_void_p(p), ctypes.c_void_p(xin.data_ptr()), n, _lib.stream_ptr(xin.device)), "copy")
    m2.run("p")
    got = torch.empty_like(want)
    _lib.check(_lib.lib().sd_copy_d2d(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(po), no, _lib.stream_ptr(got.device)), "copy")
    torch.cuda.synchronize()
    assert torch.equal(got, want) and float(want.float().mean()) != 0.0
    for name, data in (("trunc", blob[:-100]), ("long", blob + b"\0" * 8),
                       ("rec", blob[:200] + bytes([blob[200] ^ 0x40]) + blob[201:]),
                       ("weights", blob[:-5] + bytes([blob[-5] ^ 1]) + blob[-4:]),
                       ("v1", b"SDMODEL1" + blob[8:]), ("v2", b"SDMODEL2" + blob[8:])):
        bad = tmp_path / f"{name}.sdm"
        bad.write_bytes(data)
        with pytest.raises(ComaHipError, match="truncated|checksum|not a model file"):
            SdModel.load(bad, DEV)


def test_winograd_and_small_n_records_survive_save_and_load(tmp_path, hip_lib):
    """The r4 / r5 entry points are recordable: a plan of [GroupNorm + Winograd input transform] -> 16 plane products -> output transform ->
    [GroupNorm table + SiLU + 3x3 convolution with 3 output channels] is saved with its constants (the transformed weights U = G g G^T are
    computed outside the plan and must be registered as persistent when first recorded), loaded into a fresh model and replayed there:
    same bits."""
    import ctypes
    from coma_amd import _lib
    from coma_amd.sd.graph import LaunchGraph
    from coma_amd.sd.model import SdModel
    g = LaunchGraph(DEV, plan="p")
    B, H, W, C, n = 2, 16, 16, 128, 128
    gen = torch.Generator().manual_seed(4)
    r = lambda *s, k=1.0: (torch.randn(*s, generator=gen) * k).half().to(DEV)
    x, out, img = g.buf(B * H * W, C), g.buf(B * H * W, n), g.buf(B * H * W, 64, zero=True)
    w9, b9 = r(n, 9 * C, k=(9 * C) ** -0.5), r(n)
    ga, be, ga2, be2 = r(C, k=0.2) + 1, r(C, k=0.2), r(n, k=0.2) + 1, r(n, k=0.2)
    w3, b3 = r(3, 9 * n, k=(9 * n) ** -0.5), r(3)
    V = g.gn_winograd_input(ga, be, batch=B, h=H, w=W, c0=C, x0=x, eps=1e-5)
    P = g.winograd_planes(V, g.winograd_weight(w9, n=n, c=C), tiles=B * (H // 2) * (W // 2), c=C, n=n)
    g.winograd_output(P, out, batch=B, h=H, w=W, n=n, bias=b9)
    # r5: [GroupNorm table + halo-patch convolution with residual, leaving per-tile column sums] -> [table from them (rows_per_slot = 256) + ...]
    out2 = g.buf(B * H * W, n)
    wh, bh = r(n, 9 * n, k=(9 * n) ** -0.5), r(n)
    g.gn_silu_conv3x3_halo(out, ga2, be2, wh, bh, out2, batch=B, h=H, w_=W, c=n, n=n, eps=1e-6, res=out, stats=True)
    g.gn_silu_conv3x3_small_n(out2, ga2, be2, w3, b3, img, batch=B, h=H, w_=W, c=n, n=3, eps=1e-6)
    # r5: ... -> [3 -> 128 channel convolution in one launch, leaving per-tile column sums] (sd_conv3x3_c3_f16)
    from coma_amd.sd import ops
    w27, bc = r(128, 32, k=27 ** -0.5), r(128)
    w27[:, 27:] = 0
    y, csy = g.buf(B * H * W, 128), g.buf(B * H * W // 256, 2, 128, dtype=torch.float32, zero=True)
    g.add(lambda: ops.conv3x3_c3(img, w27, y, batch=B, h=H, w=W, ldx=64, bias=bc, colstats=csy), tag="conv3x3(c3)")
    g.model.bind("x", x)
    g.model.bind("img", img)
    g.model.bind("y", y)
    g.model.bind("csy", csy)
    xin = r(B * H * W, C)
    x.copy_(xin)
    g.replay()                                               # records, runs eagerly, builds the hipGraph
    g.replay()                                               # graph launch
    torch.cuda.synchronize()
    want, want_y, want_cs = img.clone(), y.clone(), csy.clone()
    assert float(want[:, :3].float().abs().max()) > 0 and float(want[:, 3:].float().abs().max()) == 0.0
    assert float(want_y.float().abs().max()) > 0 and float(want_cs.abs().max()) > 0
    path = tmp_path / "w.sdm"
    g.model.save(path)
    m2 = SdModel.load(path, DEV)
    assert m2.num_launches("p") == g.model.num_launches("p") >= 8
    px, nx = m2.binding("x")
    pi, ni = m2.binding("img")
    _lib.check(_lib.lib().sd_copy_d2d(ctypes.c_void_p(px), ctypes.c_void_p(xin.data_ptr()), nx, _lib.stream_ptr(xin.device)), "copy")
    m2.replay("p")
    got = torch.empty_like(want)
    _lib.check(_lib.lib().sd_copy_d2d(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(pi), ni, _lib.stream_ptr(got.device)), "copy")
    got_y, got_cs = torch.empty_like(want_y), torch.empty_like(want_cs)
    for name, dst in (("y", got_y), ("csy", got_cs)):
        pp, nn = m2.binding(name)
        assert nn == dst.numel() * dst.element_size()
        _lib.check(_lib.lib().sd_copy_d2d(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(pp), nn, _lib.stream_ptr(dst.device)), "copy")
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_y, want_y) and torch.equal(got_cs, want_cs)
