"""GPU: the inpainting loop end to end.
(1) fixed-mask loop at a small latent (16x16) against a torch fp32 restatement of the SAME loop (UNet oracle +
    closed-form CFG/DDIM), 3 steps, batch 2 -> latents within 5e-2 relative L2 (fp16 UNet errors compound per step);
(2) the device mask glue against a NumPy restatement (cv2.dilate 3x3 x k == (2k+1)^2 box max), bit-exact;
(3) the adaptive loop at full 512x512 / 64x64 latents with a deterministic synthetic mask plug-in: shapes, dtype,
    determinism under a seeded generator, and the adapted mask is a subset of the default mask."""
import numpy as np
import pytest
import torch

from oracle import sd_oracle as so

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_mask_adapt_matches_numpy_box_dilation(hip_lib):
    from coma_amd.sd import ops
    H = W = 64
    rng = np.random.default_rng(0)
    seg = (rng.random((H, W)) > 0.97).astype(np.uint8)
    dflt = np.zeros((H, W), np.uint8)
    dflt[8:56, 16:60] = 1
    img = rng.uniform(-1, 1, size=(3, H, W)).astype(np.float32)
    for k, use_default in ((0, False), (3, False), (20, False), (5, True)):
        mask_full = torch.empty(H, W, dtype=torch.uint8, device=DEV)
        mask_lat = torch.empty(H // 8 * W // 8, dtype=torch.float16, device=DEV)
        masked = torch.empty(H * W, 64, dtype=torch.float16, device=DEV)
        ops.mask_adapt(torch.from_numpy(seg).to(DEV), torch.from_numpy(dflt).to(DEV), torch.from_numpy(img).to(DEV), mask_full,
                       mask_lat, masked, H=H, W=W, dilate_iters=k, use_default=use_default, cpad=64)
        if use_default:
            ref = dflt.copy()
        else:
            pad = np.pad(seg, k)
            dil = np.zeros_like(seg)
            for dy in range(2 * k + 1):
                for dx in range(2 * k + 1):
                    dil |= pad[dy:dy + H, dx:dx + W]
            ref = (dil & dflt).astype(np.uint8)
        assert np.array_equal(mask_full.cpu().numpy(), ref)
        assert np.array_equal(mask_lat.float().cpu().numpy().reshape(H // 8, W // 8), ref[::8, ::8].astype(np.float32))
        m = masked.float().cpu().numpy().reshape(H, W, 64)
        exp = np.where(ref[None] > 0, 0.0, img).transpose(1, 2, 0)
        assert np.array_equal(m[:, :, :3], exp.astype(np.float16).astype(np.float32)) and float(np.abs(m[:, :, 3:]).max()) == 0.0


@pytest.mark.parametrize("B,IH,IW,steps", [(2, 128, 128, 3), (1, 512, 512, 2), (1, 192, 320, 3)])
def test_fixed_mask_loop_matches_fp32_restatement(hip_lib, B, IH, IW, steps):
    """The whole fixed-mask loop (config 2's shape in the second case: 512 x 512 image, 64 x 64 latents, CFG batch 2; the third case is a
    non-square image whose latent, 24 x 40, is ragged against every tile size) against the fp32 restatement of
    utils/adaptive_mask_inpainting.py:988-1022 step by step."""
    from coma_amd.sd import weights
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
    LH, LW = IH // 8, IW // 8
    pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=B, height=IH, width=IW, device=DEV, seed=0)
    g = torch.Generator().manual_seed(5)
    image = (torch.rand(B, 3, IH, IW, generator=g) * 2 - 1)
    mask = torch.zeros(B, 1, IH, IW)
    mask[:, :, IH // 4:3 * IH // 4, IW // 4:3 * IW // 4] = 1
    pe, ne = torch.randn(B, 77, 768, generator=g).half().float(), torch.randn(B, 77, 768, generator=g).half().float()
    lat0 = torch.randn(B, 4, LH, LW, generator=g)
    guidance = 11.0
    out = pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=steps,
               guidance_scale=guidance, latents=lat0, output_type="latent", use_adaptive_mask=False).images
    # fp32 restatement of the same loop with the pipeline's own masked-image latents (VAE parity is tested separately)
    # the restatement is evaluated by torch's fp32 kernels on the device (12 s per UNet forward on the host cores); the CPU and device
    # evaluations are tied by tests/test_sd_adaptive_gpu.py::test_restatement_on_device_equals_restatement_on_cpu
    old_tf32 = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    ustate = {k: v.to(DEV, torch.float32) for k, v in weights.random_state(weights.unet_shapes(), seed=0).items()}
    masked_lat = pipe._last_masked_lat.float().reshape(B, LH, LW, 4).permute(0, 3, 1, 2) if hasattr(pipe, "_last_masked_lat") else None
    if masked_lat is None:
        pytest.skip("pipeline does not expose masked latents")
    mask_lat = mask[:, :, ::8, ::8].to(DEV)
    alphas = so.ddim_alphas().to(DEV)
    x = lat0.clone().double().to(DEV)
    ctx = torch.cat([ne, pe]).to(DEV)
    with torch.no_grad():
        for t in so.ddim_timesteps(steps):
            inp = torch.cat([x.float(), mask_lat, masked_lat], dim=1)
            eps = so.unet_ref(ustate, torch.cat([inp, inp]), torch.full((2 * B,), float(t), device=DEV), ctx, weights.UNET_CFG)
            e = eps[:B] + guidance * (eps[B:] - eps[:B])
            x, _ = so.ddim_step_ref(e, t, x, alphas, num_inference_steps=steps)
    torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old_tf32
    x = x.cpu()
    rel = float((out.cpu().double() - x).norm() / x.norm())
    print(f"METRIC fixed-mask loop {steps} steps rel-L2 {rel:.3e}")
    assert rel <= 9e-3, rel                   # measured 4.0e-3 ... 4.3e-3 (profiles/r04_notes.md 3)


def test_adaptive_loop_full_resolution(hip_lib):
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline, SyntheticHumanMaskPredictor, default_adaptive_mask_settings
    pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=1, height=512, width=512, device=DEV, seed=0)
    pipe.register_adaptive_mask_model(SyntheticHumanMaskPredictor())
    pipe.register_adaptive_mask_settings(default_adaptive_mask_settings(50, "p"))
    rng = np.random.default_rng(0)
    image = torch.tensor(rng.uniform(-1, 1, size=(1, 3, 512, 512)).astype(np.float32))
    mask = torch.zeros(1, 1, 512, 512)
    mask[:, :, 100:420, 150:400] = 1
    pe, ne = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(1)), torch.zeros(1, 77, 768)

    def run():
        g = torch.Generator(device=DEV).manual_seed(3)
        # 6 of the 50 timesteps: strength trick keeps the schedule of the last steps
        return pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=50,
                    strength=0.12, guidance_scale=11.0, generator=g, output_type="u8", use_adaptive_mask=True,
                    enforce_full_mask_ratio=0.0, human_detection_thres=0.0).images
    a = run()
    assert tuple(a.shape) == (1, 512, 512, 3) and a.dtype == torch.uint8
    m = pipe.last_mask_image_np
    assert m is not None and m.shape == (512, 512) and set(np.unique(m)) <= {0.0, 1.0}
    assert (m <= mask[0, 0].numpy()).all(), "adapted mask must stay inside the default mask"
    b = run()
    assert torch.equal(a, b), "same generator seed -> identical image"

    class HostContract(SyntheticHumanMaskPredictor):
        """The reference's plug-in contract (uint8 HWC NumPy in, NumPy mask out, :1225-1236)."""
        accepts_device_tensor = False
        seen = []

        def __call__(self, image_u8):
            assert isinstance(image_u8, np.ndarray) and image_u8.dtype == np.uint8 and image_u8.shape == (512, 512, 3)
            HostContract.seen.append(1)
            return super().__call__(image_u8)

    pipe.register_adaptive_mask_model(HostContract())
    c = run()
    assert HostContract.seen and torch.equal(a, c), "device-tensor and NumPy plug-in paths must give the same image"
    assert np.array_equal(m, pipe.last_mask_image_np)


def test_from_pretrained_reads_a_diffusers_layout_checkpoint(tmp_path, hip_lib):
    """Write the SD-1.5-inpainting tensors under diffusers' key names as safetensors, load them through
    AdaptiveMaskInpaintPipeline.from_pretrained (the reference's DiffusionPipeline.from_pretrained call,
    src/generation/inpaint.py:64-70) and require the same image as the pipeline built from the same tensors in memory."""
    from safetensors.torch import save_file
    from coma_amd.sd import weights
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
    for sub, shapes, seed in (("unet", weights.unet_shapes(), 0), ("vae", weights.vae_shapes(), 1)):
        (tmp_path / sub).mkdir()
        st = weights.random_state(shapes, seed=seed)
        st["not_a_weight.position_ids"] = torch.arange(4)            # extra tensors are tolerated
        save_file({k: v.contiguous() for k, v in st.items()}, str(tmp_path / sub / "diffusion_pytorch_model.safetensors"))
    g = torch.Generator().manual_seed(5)
    image = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    mask = torch.zeros(1, 1, 512, 512)
    mask[:, :, 100:400, 150:380] = 1
    pe, ne = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    outs = []
    for pipe in (AdaptiveMaskInpaintPipeline.from_pretrained(str(tmp_path), batch_size=1, device=DEV),
                 AdaptiveMaskInpaintPipeline.from_random(batch_size=1, device=DEV, seed=0)):
        gen = torch.Generator(device=DEV).manual_seed(3)
        outs.append(pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2,
                         guidance_scale=7.5, generator=gen, output_type="u8", use_adaptive_mask=False).images.cpu())
        del pipe
        torch.cuda.empty_cache()
    assert outs[0].shape == (1, 512, 512, 3) and torch.equal(outs[0], outs[1])


def test_ddim_fp16_flow_is_torch_half_arithmetic(hip_lib):
    """oracle/sd_oracle.ddim_step_ref(dtype_flow="fp16") and the fp16 CFG combination of AdaptiveLoopRef, against the SAME expressions
    evaluated by torch on fp16 tensors with fp32 0-dim alphas -- what diffusers' DDIMScheduler.step / the reference's loop
    (utils/adaptive_mask_inpainting.py:1010-1017) execute in the fp16 pipeline ON THE DEVICE (the reference runs on cuda: a 0-dim CPU
    fp32 alpha enters the elementwise kernel as an fp32 scalar; torch's CPU kernels round it to fp16 first for `mul`, which is why
    this is a device test).  Bit-equal.  Nothing of libcoma_hip.so is involved: this pins the oracle's fp16 flow to torch."""
    from oracle import sd_oracle as so
    g = torch.Generator().manual_seed(3)
    dev = "cuda:0"
    alphas = so.ddim_alphas()
    for t in (981, 501, 21, 1):
        x = (torch.randn(2, 4, 8, 8, generator=g) * 1.3).to(dev)
        eu, ec = torch.randn(2, 4, 8, 8, generator=g).to(dev), torch.randn(2, 4, 8, 8, generator=g).to(dev)
        # CFG on fp16 tensors with a python-float guidance scale
        eu16, ec16 = eu.half(), ec.half()
        e16 = eu16 + 11.0 * (ec16 - eu16)
        e_ref = so.r16(so.r16(eu) + so.r16(11.0 * so.r16(so.r16(ec) - so.r16(eu))))
        assert e16.dtype == torch.float16 and torch.equal(e16.float(), e_ref)
        # scheduler step: the alphas are 0-dim fp32 tensors (scheduler.alphas_cumprod[timestep])
        a32 = alphas.float()                      # stays on the CPU, as scheduler.alphas_cumprod does
        a_t, a_p = a32[t], a32[t - 20] if t - 20 >= 0 else a32[0]
        x16 = x.half()
        x0_16 = (x16 - (1 - a_t) ** 0.5 * e16) / a_t ** 0.5
        prev16 = a_p ** 0.5 * x0_16 + (1 - a_p) ** 0.5 * e16
        assert x0_16.dtype == prev16.dtype == torch.float16
        prev, x0 = so.ddim_step_ref(e_ref, t, x, alphas.to(dev), dtype_flow="fp16")
        assert torch.equal(x0, x0_16.float()) and torch.equal(prev, prev16.float()), t
        # and the fp32 flow is the closed form
        p64, x64 = so.ddim_step_ref(e_ref, t, x, alphas.to(dev))
        assert float((prev.double() - p64).abs().max()) < 1e-2 * float(p64.abs().max()) and p64.dtype == torch.float64
