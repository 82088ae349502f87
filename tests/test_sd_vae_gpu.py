"""GPU: VAE decoder / encoder launch graphs against the torch fp32 reference of the same architecture and weights
(oracle/sd_oracle.py), full SD-1.5 widths (128/256/512/512) at a 64x64 image (8x8 latent) so the host reference is
quick.  Tolerance: relative L2 <= 3e-3, cosine >= 0.99998 = 2 x the measured 1.22e-3 ... 1.50e-3 (fp16 activations through ~30 layers)."""
import pytest
import torch

from oracle import sd_oracle as so

pytestmark = pytest.mark.gpu
REL_L2, COS = 3e-3, 0.99998
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vae_setup(hip_lib):
    from coma_amd.sd import weights
    from coma_amd.sd.vae import HipAutoencoderKL
    state = weights.random_state(weights.vae_shapes(), seed=3)
    return state, weights.VAE_CFG, HipAutoencoderKL(state, batch=2, height=64, width=64, device=DEV)


def _metrics(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    rel, cos = float((out - ref).norm() / ref.norm()), float(torch.nn.functional.cosine_similarity(out.flatten(), ref.flatten(), dim=0))
    print(f"METRIC rel-L2 {rel:.3e} cos {cos:.6f}")        # pytest -rP shows the measured distances the bars are set from
    return rel, cos


def test_decoder(vae_setup):
    state, cfg, vae = vae_setup
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0)).half().float()
    out = vae.decode(z.to(DEV), return_dict=False)[0]
    assert tuple(out.shape) == (2, 3, 64, 64)
    rel, cos = _metrics(out, so.vae_decode_ref(state, z, cfg))
    assert rel <= REL_L2 and cos >= COS, (rel, cos)


def test_encoder_moments_and_sampling(vae_setup):
    state, cfg, vae = vae_setup
    from coma_amd.sd import ops
    img = (torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().float()
    dist = vae.encode(img.to(DEV)).latent_dist
    ref = so.vae_encode_ref(state, img, cfg)                       # [2, 8, 8, 8]
    mode = dist.mode()
    rel, cos = _metrics(mode, ref[:, :4])
    assert rel <= REL_L2 and cos >= COS, (rel, cos)
    # sampling formula on the kernel's own moments: (mean + exp(0.5*clamp(logvar)) * noise) * scale
    mom = vae.enc.moments.float().reshape(2, 64, 64)[:, :, :8]
    noise = torch.randn(2, 64, 4, generator=torch.Generator().manual_seed(2))
    lat = torch.empty(2, 64, 4, device=DEV)
    ops.vae_sample(vae.enc.moments, 64, noise.to(DEV), 0.18215, 128, lat32=lat)
    exp = (mom[:, :, :4].cpu() + torch.exp(0.5 * mom[:, :, 4:].cpu().clamp(-30, 20)) * noise) * 0.18215
    assert float((lat.cpu() - exp).abs().max()) <= 1e-5 * float(exp.abs().max()) + 1e-6


@pytest.fixture(scope="module")
def vae512(hip_lib):
    from coma_amd.sd import weights
    from coma_amd.sd.vae import HipAutoencoderKL
    state = weights.random_state(weights.vae_shapes(), seed=3)
    return state, weights.VAE_CFG, HipAutoencoderKL(state, batch=1, height=512, width=512, device=DEV)


def test_decoder_512_matches_fp32_reference(vae512):
    """The benchmark's resolution: 64x64 latent -> 512x512 image.  Here M runs up to 262144 rows, so the 256 x 256,
    8-wave and 4-wave 256 x 128 tile families and the 4096-token mid-block attention are what is being compared."""
    state, cfg, vae = vae512
    # the 128-channel layers of the 512 x 512 level run as halo-patch convolutions with the GroupNorm folded in (sd_conv3x3_halo_f16)
    # (and the 256-channel ones of the 256 x 256 level, two workgroups per tile)
    assert sum("conv3x3(halo)" in tag for tag, _ in vae.dec.g.tags) == 12 and sum("conv3x3(halo)" in tag for tag, _ in vae.enc.g.tags) == 8
    assert sum("conv3x3(c3)" in tag for tag, _ in vae.enc.g.tags) == 1 and not any("im2col" in tag for tag, _ in vae.enc.g.tags)    # conv_in: one launch
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(5)).half().float()
    out = vae.decode(z.to(DEV), return_dict=False)[0]
    assert tuple(out.shape) == (1, 3, 512, 512)
    rel, cos = _metrics(out, so.vae_decode_ref(state, z, cfg))
    assert rel <= REL_L2 and cos >= COS, (rel, cos)


def test_encoder_512_matches_fp32_reference(vae512):
    state, cfg, vae = vae512
    img = (torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(6)) * 2 - 1).half().float()
    mode = vae.encode(img.to(DEV)).latent_dist.mode()
    ref = so.vae_encode_ref(state, img, cfg)
    rel, cos = _metrics(mode, ref[:, :4])
    assert rel <= REL_L2 and cos >= COS, (rel, cos)


@pytest.fixture(scope="module")
def fp32_strict():
    old = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    yield
    torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old


def test_batch_8_decoder_and_encoder_match_fp32_reference(hip_lib, fp32_strict):
    """The BENCHMARKED dispatch (batch 8 at 512 x 512: the 8-wave wide-head mid-block attention with 128 queries per workgroup, n / 128 halo
    workgroups per tile at every level that fills the chip, the 256 x 256 GEMM tiles) at model level: images 0 and 7 of a batch of 8 distinct
    inputs against vae_decode_ref / vae_encode_ref (utils/adaptive_mask_inpainting.py:1111-1115, :675-684), the restatement evaluated by
    torch's fp32 device kernels one image at a time (the images of a batch are independent).  Same bars as batch 1."""
    from coma_amd.sd import weights
    from coma_amd.sd.vae import HipAutoencoderKL
    state = weights.random_state(weights.vae_shapes(), seed=3)
    cfg = weights.VAE_CFG
    vae = HipAutoencoderKL(state, batch=8, height=512, width=512, device=DEV)
    assert any("attention(wide)" in tag for tag, _ in vae.dec.g.tags) and sum("conv3x3(halo)" in tag for tag, _ in vae.dec.g.tags) >= 12
    dstate = {k: v.to(DEV) for k, v in state.items()}
    z = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(15)).half().float()
    out = vae.decode(z.to(DEV), return_dict=False)[0].clone()
    assert tuple(out.shape) == (8, 3, 512, 512)
    img = (torch.rand(8, 3, 512, 512, generator=torch.Generator().manual_seed(16)) * 2 - 1).half().float()
    mode = vae.encode(img.to(DEV)).latent_dist.mode().clone()
    for b in (0, 7):
        rel, cos = _metrics(out[b:b + 1], so.vae_decode_ref(dstate, z[b:b + 1].to(DEV), cfg))
        assert rel <= REL_L2 and cos >= COS, ("decode", b, rel, cos)
        rel, cos = _metrics(mode[b:b + 1], so.vae_encode_ref(dstate, img[b:b + 1].to(DEV), cfg)[:, :4])
        assert rel <= REL_L2 and cos >= COS, ("encode", b, rel, cos)
    assert float((out[0] - out[7]).abs().max()) > 0.1                 # distinct images came out distinct


def test_halo_convolutions_agree_with_the_groupnorm_plus_gemm_graph(vae512):
    """The r5 decoder / encoder (every ResNet convolution that fills the chip a halo-patch convolution with the GroupNorm + SiLU applied on the way into LDS) against
    the SAME networks built the r4 way (GroupNorm kernel -> implicit GEMM): two launch lists computing the same products with the same
    fp16 storage points up to accumulation order: each is 1.3e-3 ... 1.5e-3 from the fp32 restatement, and they are 0.8e-3 (decoder) / 1.7e-3
    (encoder moments) from each other -- bar at 2 x that."""
    from coma_amd.sd import vae as vae_mod
    from coma_amd.sd.vae import HipAutoencoderKL
    state, cfg, vae = vae512
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(7)).half().float()
    img = (torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(8)) * 2 - 1).half().float()
    out_h = vae.decode(z.to(DEV), return_dict=False)[0].clone()
    mom_h = vae.encode(img.to(DEV)).latent_dist.mode().clone()
    try:
        vae_mod._VaeBase.halo_conv = False
        old = HipAutoencoderKL(state, batch=1, height=512, width=512, device=DEV)
    finally:
        vae_mod._VaeBase.halo_conv = True
    assert not any("halo" in tag for tag, _ in old.dec.g.tags + old.enc.g.tags) and any("halo" in tag for tag, _ in vae.dec.g.tags)
    out_g = old.decode(z.to(DEV), return_dict=False)[0]
    mom_g = old.encode(img.to(DEV)).latent_dist.mode()
    for a, b in ((out_h, out_g), (mom_h, mom_g)):
        rel, cos = _metrics(a, b)
        assert rel <= 3.5e-3 and cos >= 0.99998, (rel, cos)
