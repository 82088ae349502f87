#!/usr/bin/env python3
"""bench.py -- throughput of ComA's dense hot path on MI355X (contract: see the round brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload inpaint|contact]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for N > 1: started as a plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) the script re-executes
itself under torch.distributed.run with one rank per GPU (the fan-out the reference does with one process per GPU,
scripts/generation/inpaint.sh:205-268 of the reference); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

Primary workload (`metric` / `value`, BASELINE.json config 2): SD-1.5 inpaint 512 x 512, 50 DDIM steps, batch 8 images per GPU
(UNet batch 16 with classifier-free guidance), fixed mask.  One "step" = one whole image batch with inputs resident in HBM:
masked-image VAE encode, 50 x (UNet hipGraph + CFG / DDIM kernel), VAE decode to uint8.  `value` = images/s summed over all ranks
(weak scaling: independent image batches, no data-path collective), time = MAX over ranks between two barrier + synchronize pairs.
Further sections on the same line, every rank taking part (only rank 0 prints):
  secondary        ComA K1-K3 accumulation at config 4's per-GPU slice (H=10475 x O=180 x N=250 bins, 64 samples per step), K steps and
                   then ONE all-reduce(SUM) of the state when N > 1 (timed inside, reported as final_allreduce_ms)
  occupancy        config 5's per-GPU row share (H=1310, R=128, S=2000) through the fused pass + all-reduce(MAX) when N > 1
  adaptive_loop    config 3's shape: the full adaptive-mask loop (49 steps, 21 re-estimations) at batch 8, synthetic mask plug-in
  adaptive_loop_b1 the same, one image per pipeline call (the reference's own call shape)
  adaptive_loop_pointrend  the same at batch 8 with the PointRend-ARCHITECTURE plug-in on the device (coma_amd/seg, fp32; seeded random weights,
                   4 detections per image forced) + that network's own forward time and fp32-MFMA roofline fraction
  roofline         conv / linear GEMM family of one UNet forward: algorithmic flops / HIP-event time against the 2.5 PF dense MFMA peak
  cpu_baseline     the oracle restatements timed on the host cores (N = 1 only; bounded samples, stated)
`--workload contact` swaps primary and secondary.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 vector
HBM_PEAK_GBS = 8000.0


def flop_per_pair(N):
    """Algorithmic flop per (sample, h, o) pair, SURVEY.md 8d: 14 (distance/proximity) + 2*43 (canonicalise)
    + 13 per bin per grid (5 dot, 2 clip, acos, square, /sigma^2, exp, recip, add)."""
    return 100 + 26 * N


def cpu_baseline(seconds_budget=12.0):
    """SURVEY.md 8d, cfg 1 (H=1000, O=180, N=250) on the host cores, two ports of the reference's CPU path with its dtype flow:
    `value` = the torch-CPU restatement on ALL host threads (what the reference itself runs with device="cpu"), and under
    `single_thread_numpy` the NumPy oracle the parity tests use (one thread).  Bounded samples."""
    from oracle import coma_oracle as orc
    from oracle import coma_oracle_torch as ot
    from tests.synth import cfg1_samples
    H, O, N = 1000, 180, 250
    samples = cfg1_samples(9, seed=0, H=H, O=O)

    def run(m):
        m.aggregate_sample(**samples[0])           # warm-up
        done, t0 = 0, time.perf_counter()
        for s in samples[1:]:
            m.aggregate_sample(**s)
            done += 1
            if time.perf_counter() - t0 > seconds_budget:
                break
        return done, time.perf_counter() - t0
    nt = torch.get_num_threads()
    d_t, t_t = run(ot.ComATorch(H, O, N, 0.07, 0.03, sigma=0.25, eps=1e-10))
    d_n, t_n = run(orc.ComAOracle(H, O, N, 0.07, 0.03, sigma=0.25, eps=1e-10))
    return {"value": d_t * H * O / t_t, "unit": "vertex-pair contacts/s", "cores": nt, "kind": "port",
            "sample": f"{d_t} samples of config 1 (H=1000,O=180,N=250) through oracle/coma_oracle_torch.py (torch CPU, {nt} threads, f64 "
                      f"scores added into f32 grids as in utils/coma.py:279-323), {t_t:.1f} s; host has {os.cpu_count()} logical cores",
            "single_thread_numpy": {"value": d_n * H * O / t_n, "unit": "vertex-pair contacts/s", "cores": 1, "kind": "port",
                                    "sample": f"{d_n} samples through oracle/coma_oracle.py (NumPy), {t_n:.1f} s"}}


MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16/bf16 MFMA


def cpu_baseline_inpaint():
    """torch fp32 CPU restatement of the UNet (oracle/sd_oracle.py) on the host cores: ONE forward at batch 2
    (cond + uncond of one image, 64x64 latents), extrapolated to 50 steps; VAE encode/decode (~4 % of the flops) ignored."""
    from coma_amd.sd import weights
    from oracle import sd_oracle as so
    state = weights.random_state(weights.unet_shapes(), seed=0)
    g = torch.Generator().manual_seed(0)
    x, ctx, t = torch.randn(2, 9, 64, 64, generator=g), torch.randn(2, 77, 768, generator=g), torch.tensor([961.0, 961.0])
    with torch.no_grad():
        t0 = time.perf_counter()
        so.unet_ref(state, x, t, ctx, weights.UNET_CFG)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / (50 * dt), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 UNet forward at batch 2 (one image, cond+uncond, 64x64 latents) through oracle/sd_oracle.py (torch fp32 "
                      f"CPU, {torch.get_num_threads()} threads): {dt:.2f} s, EXTRAPOLATED x50 steps; VAE ignored"}


def bench_inpaint(args, dev, world, rank):
    """BASELINE.json config 2: SD-1.5 inpaint 512x512, 50 DDIM steps, batch 8, fixed mask, CFG 11 (UNet batch 16),
    seeded random fp16 weights of the SD-1.5-inpainting shapes, synthetic prompt embeddings / image / mask.  One step =
    one whole batch: masked-image VAE encode, 50 x (UNet graph + CFG/DDIM kernel), VAE decode to uint8."""
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
    B = args.images
    pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=B, height=512, width=512, device=dev, seed=0,
                                                   use_graph=not args.eager)
    g = torch.Generator().manual_seed(100 + rank)
    image = torch.rand(B, 3, 512, 512, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, 512, 512)
    mask[:, :, 128:384, 128:384] = 1                       # centred 256^2 box (cfg 2)
    pe, ne = torch.randn(B, 77, 768, generator=g), torch.randn(B, 77, 768, generator=g)
    # the contract's "inputs already resident in HBM when the timed region starts": image, mask and prompt embeddings live on the device (handing
    # over host tensors costs 2.5-3 ms per batch of 8, 0.3 %: scripts/time_pipeline.py, "host inputs" against "device inputs")
    image, mask, pe, ne = (t.to(dev) for t in (image, mask, pe, ne))
    gens = torch.Generator(device=dev)

    def step(seed):
        gens.manual_seed(seed)
        return pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=args.ddim_steps,
                    strength=1.0, guidance_scale=11.0, generator=gens, output_type="u8", use_adaptive_mask=False).images

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(1000 + k)
    barrier()
    dt = time.perf_counter() - t0
    assert out.shape == (B, 512, 512, 3) and out.dtype == torch.uint8
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    res = None
    if rank == 0:
        # roofline of the dominant kernel (the implicit-GEMM conv/linear kernel): per-launch HIP-event timing of the
        # UNet launch list, outside the timed region, same process
        prof = pipe.unet.g.profile(reps=2)
        # the conv / linear family: every GEMM launch; a Winograd convolution (deep ResNet levels) counts with the ALGORITHMIC flops of the
        # 3x3 convolution it computes and with the time of all of its launches (transforms + plane products) -- `executed` is what the MFMAs issued
        fam = lambda tag: tag.startswith("gemm") or "winograd" in tag
        gemm = [(fl, ms) for tag, fl, ms in prof if fam(tag)]
        exec_fl = sum(e for (tag, _), e in zip(pipe.unet.g.tags, pipe.unet.g.exec_tags) if fam(tag))
        n_gemm = sum(1 for tag, _, _ in prof if tag.startswith("gemm"))
        alg_bytes = sum(b for (tag, _), b in zip(pipe.unet.g.tags, pipe.unet.g.alg_bytes) if tag.startswith("gemm"))     # conv_gemm launches only, like `traffic`
        attn = [(fl, ms) for tag, fl, ms in prof if tag.startswith("attention")]
        tot_ms = sum(ms for _, _, ms in prof)
        g_fl, g_ms = sum(f for f, _ in gemm), sum(m for _, m in gemm)
        flops_img = (pipe.unet.g.flops * 50 + pipe.vae.dec.g.flops + pipe.vae.enc.g.flops) / B
        vendor = vendor_gemm_reference(dev)
        # the two VAE graphs on their own (outside the timed region): 1 + 1 of them per image in this workload, 21 + 21 per image in the
        # adaptive loop, where they are ~ 43 % of the time
        vae_ms = {}
        for name, run in (("decode_ms", pipe.vae.dec.decode_static), ("encode_ms", pipe.vae.enc.encode_static)):
            run()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                run()
            e.record()
            torch.cuda.synchronize()
            vae_ms[name] = a.elapsed_time(e) / 3
        value = world * B * args.steps / dt
        res = {
            "metric": "HOI images/sec (50-step SD-1.5 inpaint, 512x512, fixed mask, CFG; BASELINE metric part 1)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"SD-1.5-inpaint 512x512, 50 DDIM steps, batch {B} images per GPU per step (UNet batch {2 * B}), "
                                   "fixed centred 256^2 mask, guidance 11, 1 VAE encode + 1 VAE decode per image "
                                   "(BASELINE.json config 2); random-init weights of the SD-1.5-inpainting architecture",
                       "parallelism": f"independent image batches on {world} GPU(s), no collective"},
            "tflop_per_image": flops_img / 1e12, "achieved_tflops_whole_loop": value / world * flops_img / 1e12,
            "roofline": {"bound": "mfma", "kernel": "sd::conv_gemm_kernel<WM,WN,TN,BK,STAGES> (all conv3x3/1x1/linear launches of one UNet forward)",
                         "achieved": g_fl / g_ms / 1e9, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": g_fl / g_ms / 1e9 / MFMA_F16_PEAK_TFLOPS, "launches": n_gemm,
                         "avg_launch_ms": g_ms / n_gemm, "flops_per_forward": g_fl,
                         "executed_tflops": exec_fl / g_ms / 1e9, "executed_flops_per_forward": exec_fl,
                         "executed_frac": exec_fl / g_ms / 1e9 / MFMA_F16_PEAK_TFLOPS,      # what the MFMAs issued against the same peak
                         "flops_what": "algorithmic: 2*M*N*K of every conv3x3 / 1x1 / linear operator of a UNet forward; the 24 ResNet convolutions of "
                                       "the 16x16 / 8x8 levels run as Winograd F(2x2,3x3) (16 plane products = 4/9 of the multiplies) and are timed "
                                       "with their transform launches; executed_* counts the MFMA flops actually issued",
                         "algorithmic_bytes": alg_bytes / n_gemm, "algorithmic_bytes_what": "per launch, mean over the launch list: every input "
                         "activation, weight, bias / residual tile read once + the output written once (fp16)",
                         "traffic": UNET_GEMM_PMC_TRAFFIC_BYTES if B == 8 else None,
                         "traffic_source": "profiles/r05_unet_gemm_traffic.txt: (2*FETCH_SIZE + WRITE_SIZE) per conv_gemm launch, mean over "
                                           "the launches of eager UNet forwards at batch 16 (fabric-side requests of the 8 L2s: "
                                           "weights are pulled once per XCD and mostly hit the Infinity Cache)",
                         "unet_forward_ms_eager_sum": tot_ms,
                         "vendor_gemm": vendor,
                         "attention": {"achieved": sum(f for f, _ in attn) / sum(m for _, m in attn) / 1e9, "unit": "TFLOP/s"}},
            "vae": dict(vae_ms, batch=B, what="VAE decoder / encoder hipGraph at 512 x 512, HIP events, mean of 3 replays "
                                              "(every ResNet convolution = halo-patch convolution with the GroupNorm + SiLU folded in)"),
        }
    del pipe
    torch.cuda.empty_cache()
    return res


def vendor_gemm_reference(dev, n=8192, seconds=1.0):
    """What the 1400 W package cap leaves of the nominal MFMA peak on THIS box: torch.matmul (hipBLASLt) on N(0,1) fp16 operands,
    n^3, timed for about a second after a warm-up.  A reference point next to `roofline.achieved`; never part of
    the product path or of the timed region."""
    a = torch.randn(n, n, device=dev).half()
    b = torch.randn(n, n, device=dev).half()
    for _ in range(20):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        reps += 20
    dt = (time.perf_counter() - t0) / reps
    return {"achieved": 2 * n ** 3 / dt / 1e12, "unit": "TFLOP/s", "what": f"torch.matmul (hipBLASLt) {n}^3 fp16, N(0,1) operands, same process"}


def bench_contact(args, dev, world, rank):
    from utils.coma import ComA
    H, O, N, S = args.human_res, args.obj_res, args.normal_res, args.samples
    coma = ComA(H, O, N, 0, proximity_settings=dict(spatial_grid_size=0.07, spatial_grid_thres=0.03),
                normal_gaussian_sigma=0.25, eps=1e-10, device=dev)
    # synthetic inputs of config 4's shape, generated on the device (resident in HBM before the timed region)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    lo = torch.tensor([-0.3, -0.15, -0.85], device=dev)
    hi = torch.tensor([0.3, 0.15, 0.85], device=dev)
    hv = lo + (hi - lo) * torch.rand([S, H, 3], generator=g, device=dev)
    hn = torch.nn.functional.normalize(torch.randn([S, H, 3], generator=g, device=dev), dim=-1)
    g0 = torch.Generator(device=dev).manual_seed(99)           # the object is the same on every rank
    on = torch.nn.functional.normalize(torch.randn([O, 3], generator=g0, device=dev), dim=-1)
    ov = on * 0.2 + torch.tensor([0.0, -0.15, 0.3], device=dev)
    steps, warmup = args.contact_steps, 2

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        coma.accumulate_device(hv, hn, ov, on)
    if world > 1:
        coma.all_reduce()                           # warm-up of the collective (communicator set-up is not part of the job)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        coma.accumulate_device(hv, hn, ov, on)      # same stream as the events (torch current stream)
        ev[i][1].record()
    t_ar = 0.0
    if world > 1:
        # north_star: the collective runs ONCE, on the final affordance histogram -- K accumulation steps, then one
        # all-reduce(SUM) of the state; it is inside the timed region and also reported on its own
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        coma.all_reduce()
        torch.cuda.synchronize()
        t_ar = time.perf_counter() - t1
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return None
    flops = S * H * O * flop_per_pair(N)
    achieved = flops / (kern_ms * 1e-3) / 1e12
    alg_bytes = 24 * (S * H + O) + (16 * N + 24) * H * O
    return {
        "metric": "vertex-pair contacts/sec (ComA K1-K3 accumulation; BASELINE metric part 2)",
        "value": world * S * H * O * steps / dt, "unit": "vertex-pair contacts/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "final_allreduce_ms": t_ar * 1e3 if world > 1 else None,
        "config": {"workload": f"ComA contact+orientation accumulation, H={H} SMPL-X verts x O={O} object points x "
                               f"N={N} bins, {S} samples per GPU per step (BASELINE.json config 4 per-GPU slice)"
                               + (f"; ONE RCCL all-reduce(SUM) of the ComA state after the {steps} steps, inside the timed region" if world > 1 else ""),
                   "parallelism": f"samples sharded over {world} GPU(s)"},
        "roofline": {"bound": "fp32_valu", "kernel": "coma::contact_accumulate_kernel", "achieved": achieved,
                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
                     "kernel_ms": kern_ms, "flop_per_pair": flop_per_pair(N), "algorithmic_hbm_bytes": alg_bytes,
                     "traffic": CONTACT_PMC_TRAFFIC_BYTES if (S, H, O, N) == (64, 10475, 180, 250) else None,
                     "traffic_source": "profiles/r05_inpaint_pmc.txt (contact_accumulate_kernel): (2*FETCH_SIZE + WRITE_SIZE) KiB per launch"},
    }


FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (= the fp32 vector rate)


def pointrend_plugin(dev, batch, forced=4):
    """The PointRend-architecture mask plug-in on the device (coma_amd/seg) with SEEDED RANDOM weights (the checkpoint cannot be fetched) and
    FORCED detections: random class scores put the 100-detection cap of the reference's config on every image, and the mask head's cost is
    proportional to the detections -- so the cap (TEST.DETECTIONS_PER_IMAGE) is set to `forced` = 4, the order of what a real checkpoint finds
    in an HOI render: every image carries exactly 4 instances through the mask head.  -> (plug-in, plan, record)."""
    from coma_amd.seg import weights as SW
    from coma_amd.seg.predictor import HipPointRendPredictor
    state = SW.random_state(seed=0, cls_gain=0.2, delta_gain=0.1, person_bias=3.0)
    pred = HipPointRendPredictor(pointrend_thres=0.2, device=dev, state=state, detections_per_image=forced)
    plan = pred.pointrend_seg_model.plan(batch, 512, 512)
    g = torch.Generator().manual_seed(9)
    low = torch.rand(batch, 3, 16, 16, generator=g)
    img = (torch.nn.functional.interpolate(low, size=(512, 512), mode="bicubic").clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(dev)
    mean = float(plan(img)["count"].float().mean())
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        plan(img)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 3
    prof = plan.g.profile(reps=2)
    gemm = [(fl, t) for tag, fl, t in prof if tag.startswith("seg gemm")]
    big = [(fl, t) for fl, t in gemm if fl >= 2e9 * batch]          # the backbone / FPN / RPN / box-head convolutions (fixed work per image)
    rec = {"detections_cap": forced, "detections_per_image_on_a_noise_image": mean, "forward_ms": ms, "batch": batch,
           "launches": len(plan.g.launches), "gflop_per_image_at_capacity": plan.g.flops / batch / 1e9,
           "roofline": {"bound": "mfma", "kernel": "seg::conv_gemm_f32_kernel (v_mfma_f32_32x32x2_f32; layers of >= 2 GFLOP per image)",
                        "achieved": sum(f for f, _ in big) / sum(t for _, t in big) / 1e9, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": sum(f for f, _ in big) / sum(t for _, t in big) / 1e9 / FP32_MFMA_PEAK_TFLOPS, "launches": len(big),
                        "all_gemm_launches_ms": sum(t for _, t in gemm), "eager_sum_ms": sum(t for _, _, t in prof),
                        "algorithmic_bytes": sum(b for (tag, _), b in zip(plan.g.tags, plan.g.alg_bytes) if tag.startswith("seg gemm")) / max(1, len(gemm)),
                        "traffic": SEG_GEMM_PMC_TRAFFIC_BYTES if (batch, forced) == (8, 4) else None,
                        "traffic_source": "profiles/r06_seg_pmc.txt: (2*FETCH_SIZE + WRITE_SIZE) over the five conv_gemm_f32 instantiations of one eager "
                                          "forward at batch 8 (15.6 GB fetched, 6.4 GB written) / 90 launches; a CONSTANT from the builder's counter pass, "
                                          "not measured by this run",
                        "note": "fp32 as the reference runs detectron2 (no autocast): priced against the fp32 matrix peak; the point-head / "
                                "coarse-head GEMMs are gated by the device-side detection count and are not in `achieved`"}}
    return pred, plan, rec


ADAPTIVE_SECTIONS = {            # name -> (images per call or None = --images, timed loops, plug-in)
    "adaptive_b8": (None, 2, "synthetic"),
    "adaptive_b1": (1, 3, "synthetic"),        # ONE image per pipeline call: the reference's own call shape (src/generation/inpaint.py:280-352 there)
    "pointrend": (None, 2, "pointrend"),       # the PointRend-architecture plug-in on the device (labelled: random weights, forced detections)
}
CHILD_TIMEOUT_S = 300


def adaptive_measure(dev, images, n, plugin, seed_rank=0):
    """BASELINE.json config 3 shape: the full adaptive-mask loop on a batch of independent 512x512 images (the reference runs
    one image per call; here each image of the batch adapts its own mask), strength 0.98 -> 49 steps, 21 mask re-estimations
    (x0 decode + mask plug-in + device mask glue + VAE re-encode), the plug-in inside the timed region.  -> (seconds per loop, seg record)"""
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline, SyntheticHumanMaskPredictor, default_adaptive_mask_settings
    AB = images
    pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=AB, height=512, width=512, device=dev, seed=0)
    seg_rec = seg_plan = None
    if plugin == "pointrend":
        model, seg_plan, seg_rec = pointrend_plugin(dev, AB)
        pipe.register_adaptive_mask_model(model)
    else:
        pipe.register_adaptive_mask_model(SyntheticHumanMaskPredictor())
    pipe.register_adaptive_mask_settings(default_adaptive_mask_settings(50, "p"))
    g = torch.Generator().manual_seed(5 + seed_rank)
    image = torch.rand(AB, 3, 512, 512, generator=g) * 2 - 1
    mask = torch.zeros(AB, 1, 512, 512)
    mask[:, :, 100:420, 150:400] = 1
    pe, ne = torch.randn(AB, 77, 768, generator=g), torch.randn(AB, 77, 768, generator=g)
    image, mask, pe, ne = (t.to(dev) for t in (image, mask, pe, ne))         # inputs resident in HBM, as for the primary workload
    gen = torch.Generator(device=dev)

    def one(seed):
        gen.manual_seed(seed)
        return pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=50,
                    strength=0.98, guidance_scale=11.0, generator=gen, output_type="u8", use_adaptive_mask=True,
                    enforce_full_mask_ratio=0.0, human_detection_thres=0.015).images

    one(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        one(1 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    if seg_rec is not None:
        seg_rec["detections_per_image_after_the_last_loop"] = float(seg_plan.out["count"].float().mean())
    return dt, seg_rec


def child_main(name, device_index, images, seed_rank):
    """`bench.py --child-section NAME`: one adaptive section measured in its OWN process, one JSON line on stdout."""
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    im, n, plugin = ADAPTIVE_SECTIONS[name]
    dt, seg_rec = adaptive_measure(dev, im if im is not None else images, n, plugin, seed_rank)
    print(json.dumps({"child_section": name, "s_per_loop": dt, "segmentation": seg_rec}), flush=True)


def bench_adaptive_isolated(name, args, dev, world, rank):
    """One adaptive section in a CHILD process on this rank's device (no collective inside: the loops of different ranks are independent;
    the parent takes the MAX of the ranks' times).  A device fault or a hang in an optional section -- the seg plan's captured graph hung
    about one process in three before its memset node was replaced, profiles/r06_notes.md 2 -- then costs that section, not the line."""
    import subprocess
    im, n, plugin = ADAPTIVE_SECTIONS[name]
    AB = im if im is not None else args.images
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.abspath(__file__), "--child-section", name, "--child-device", str(dev.index), "--images", str(args.images),
           "--child-seed-rank", str(rank)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    err, rec, dt = None, None, float("inf")
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=CHILD_TIMEOUT_S, env=env, start_new_session=True)
        lines = [l for l in p.stdout.splitlines() if l.startswith('{"child_section"')]
        if p.returncode == 0 and lines:
            rec = json.loads(lines[-1])
            dt = rec["s_per_loop"]
        else:
            err = f"child exited with {p.returncode}: {(p.stderr or '').strip().splitlines()[-3:]}"
    except subprocess.TimeoutExpired:
        err = f"child exceeded {CHILD_TIMEOUT_S} s and was killed"
    t = torch.tensor([dt if err is None else 1e30], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank != 0:
        return None
    if float(t.item()) >= 1e30:
        return {"error": err or "a rank's child process failed", "plugin": plugin}
    dt = float(t.item()) / AB
    what = ("PointRend-ARCHITECTURE plug-in on the device (coma_amd/seg: fp32 R50-FPN + RPN + box head + PointRend point head), seeded RANDOM "
            "weights, 4 detections per image FORCED (TEST.DETECTIONS_PER_IMAGE = 4)" if plugin == "pointrend" else "synthetic mask plug-in")
    out = {"metric": "adaptive-mask HOI images/s (49 steps, 21 mask re-estimations)", "value": world / dt, "unit": "images/s", "n_gpus": world,
           "s_per_image": dt, "plugin": plugin, "process": "child (isolated from the primary measurement)",
           "config": {"workload": f"config 3 shape: full adaptive loop, {what}, 512x512, {AB} images per call per GPU"}}
    if rec.get("segmentation") is not None:
        out["segmentation"] = rec["segmentation"]
    return out


def bench_adaptive(args, dev, world, rank):
    return bench_adaptive_isolated("adaptive_b8", args, dev, world, rank)


def bench_adaptive_b1(args, dev, world, rank):
    return bench_adaptive_isolated("adaptive_b1", args, dev, world, rank)


def bench_adaptive_pointrend(args, dev, world, rank):
    return bench_adaptive_isolated("pointrend", args, dev, world, rank)


def bench_occupancy(args, dev, world, rank):
    """BASELINE.json config 5, one GPU's share: human-vertex rows are sharded over the ranks of an 8-GPU job (1310 of 10475 rows
    each; weak scaling: every rank of this run holds such a share), every rank runs all S=2000 samples through the fused pass over
    its [1310, 128^3] grid (11 GB), and the [R,R,R] field is MAX-all-reduced (NaN-propagating) when there is more than one rank."""
    from coma_amd import dist as cdist
    from utils.coma_occupancy import ComA_Occupancy
    H, R, S = 1310, 128, 2000
    occ = ComA_Occupancy(scale_tolerance=3.0, human_res=H, obj_res=1, normal_res=0, spatial_res=R, device=dev)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    q = (torch.rand([S, H, 3], generator=g, device=dev) * 2.6 - 1.3).contiguous()      # ~21 % of the points fall outside the grid

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    occ.accumulate_device(q[:8])
    out = occ.return_aggregated_spatial_grids()          # warm-up of the fused pass
    if world > 1:
        cdist.all_reduce_max_nan(out)                    # ... and of the collective
    reps, ms_kern = 3, []
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        occ.reset()
        a, b = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        a.record()
        occ.accumulate_device(q)                         # staged only: the reducer below runs splat + row sums + max as one pass
        out = occ.return_aggregated_spatial_grids()
        b.record()
        if world > 1:
            cdist.all_reduce_max_nan(out)
        ms_kern.append((a, b))
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ms_step = float(t.item()) / reps * 1e3               # includes reset() (its 11 GB memset is deferred and never needed here) and the collective
    ms = float(np.mean([a.elapsed_time(b) for a, b in ms_kern]))
    # the unfused route (global-atomic splat, row sums, normalise + max: three passes over the grid) for comparison
    occ.reset()
    c, d = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    c.record()
    occ.accumulate_device(q, lazy=False)
    occ.return_aggregated_spatial_grids()
    d.record()
    torch.cuda.synchronize()
    ms_unfused = c.elapsed_time(d)
    del occ
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    # algorithmic bytes of SURVEY.md 8d structure B: the per-vertex grid written once + the samples re-read once per x-plane slab;
    # hard floor: the grid written once and nothing else
    slabs = R // max(1, 20480 // (R * R)) if R * R <= 20480 else R
    alg = 4 * H * R**3 + 12 * S * H * slabs
    floor = 4 * H * R**3
    return {"metric": "occupancy splats/s (ComA_Occupancy K5+K6 fused, R=128)", "value": world * S * H / (ms * 1e-3), "unit": "splats/s",
            "n_gpus": world, "dense_equivalent_voxel_tests_per_s": world * S * H * R**3 / (ms * 1e-3), "fused_ms": ms, "unfused_ms": ms_unfused,
            "ms_per_step_with_reset_and_collective": ms_step,
            "config": {"workload": f"H={H} rows/GPU (10475/8), R={R}, S={S}, scale_tolerance 3 (config 5 per-GPU share): splat, row sums, "
                                   "max over humans in one pass, raw per-vertex grid left in HBM"
                                   + ("; NaN-propagating all-reduce(MAX) of the [R,R,R] field per step" if world > 1 else ""),
                       "parallelism": f"rows sharded, {world} rank(s) each holding a 1310-row share"},
            # frac is quoted on the HARD FLOOR (the [H, R^3] f32 grid written once, nothing else): the structure-B formula's second term
            # (samples re-read once per x-plane slab) never reaches HBM -- the incidences are bucketed once and read once -- so counting it
            # would flatter the number; it stays as a secondary key
            "roofline": {"bound": "hbm", "kernel": "coma::occupancy_fused_kernel (+ rowprep, groupmax)", "achieved": floor / (ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": floor / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes": floor, "structure_b_bytes": alg, "structure_b_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": OCCUPANCY_PMC_TRAFFIC_BYTES, "traffic_source": OCCUPANCY_PMC_SOURCE}}


# HBM-side bytes per launch from the committed rocprofv3 PMC passes of this round (separate --pmc passes for FETCH_SIZE and WRITE_SIZE;
# FETCH_SIZE doubled: gfx950 reports half of a coalesced stream, MI355X_MICROARCH.md):
#   profiles/r06_seg_pmc.txt             (2*FETCH + WRITE) of the seg::conv_gemm_f32_kernel instantiations, one eager forward at batch 8, / 90 launches
#   profiles/r05_unet_gemm_traffic.txt   (2*FETCH + WRITE) / conv_gemm launch, mean over eager UNet forwards at batch 16
#   profiles/r05_inpaint_pmc.txt         contact_accumulate_kernel: FETCH_SIZE 1.91569e6 KiB, WRITE_SIZE 3.71278e6 KiB per launch
#   profiles/r05_inpaint_pmc.txt         occupancy at the config-5 share: fused WRITE 11.02 GB + 2 * FETCH 0.124 GB, rowprep 2 * 0.209 + 0.232 GB,
#                                        groupmax 0.04 GB
UNET_GEMM_PMC_TRAFFIC_BYTES = int(158.17e6)
SEG_GEMM_PMC_TRAFFIC_BYTES = int(21.92e9 / 90)   # 243.6 MB per launch against 177 MB algorithmic (1.37 x): halo re-reads of the 3 x 3 layers across XCDs + split-K slabs
OCCUPANCY_PMC_TRAFFIC_BYTES = int(11.96e9)   # fused 11.02 + 0.25, rowprep 0.42 + 0.23, groupmax 0.04 GB
OCCUPANCY_PMC_SOURCE = ("profiles/r05_inpaint_pmc.txt: (2*FETCH_SIZE + WRITE_SIZE) of occupancy_rowprep (0.42 GB fetched, 0.23 GB of bucketed 16-byte "
                        "incidences written) + occupancy_fused (11.02 GB written, 0.25 GB fetched) + occupancy_groupmax (0.04 GB) at H=1310, R=128, S=2000: "
                        "the grid written once + the incidences; the 4 GB 'samples re-read per slab' term of the structure-B formula never reaches HBM")
CONTACT_PMC_TRAFFIC_BYTES = int((2 * 1.91569e6 + 3.71278e6) * 1024)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed steps of the primary workload (one step = one image batch)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="inpaint", choices=["inpaint", "contact"],
                    help="primary workload (the JSON line's metric/value); the other one is reported under 'secondary'")
    ap.add_argument("--images", type=int, default=8, help="images per GPU per step (inpaint)")
    ap.add_argument("--samples", type=int, default=64, help="samples per GPU per step (contact)")
    ap.add_argument("--contact-steps", type=int, default=10)
    ap.add_argument("--human-res", type=int, default=10475)
    ap.add_argument("--obj-res", type=int, default=180)
    ap.add_argument("--normal-res", type=int, default=250)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--skip", default="", help="comma-separated sections to leave out: occupancy, adaptive (profiling runs: rocprofv3 --pmc "
                    "segfaults under hipGraph replay on this image, and the adaptive loop always replays graphs)")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the hipGraph "
                    "(profiling aid: rocprofv3 --pmc segfaults under graph replay on this image)")
    ap.add_argument("--ddim-steps", type=int, default=50, help="only for profiling runs; the metric is defined at 50")
    ap.add_argument("--child-section", default=None, choices=sorted(ADAPTIVE_SECTIONS), help="internal: measure one adaptive section in this process")
    ap.add_argument("--child-device", type=int, default=0)
    ap.add_argument("--child-seed-rank", type=int, default=0)
    args = ap.parse_args()
    if args.child_section:
        return child_main(args.child_section, args.child_device, args.images, args.child_seed_rank)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on a free local port
        # (the form the driver uses for N = 1 must also work for N > 1); the children see WORLD_SIZE and take the branch below
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("COMA_BENCH_PRINT_LAUNCH") == "1":       # CPU test hook: show the launcher command instead of running it
            print(json.dumps(cmd))
            return
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # COMA_BENCH_SHARED_DEVICE=1: control-flow test of the N>1 path on a 1-GPU box (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device); never set by the driver
    shared = os.environ.get("COMA_BENCH_SHARED_DEVICE") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "contact":
        args.contact_steps = args.steps
    inp = None if (args.workload == "contact" and args.no_secondary) else bench_inpaint(args, dev, world, rank)
    con = None if (args.workload == "inpaint" and args.no_secondary) else bench_contact(args, dev, world, rank)
    occ = ada = ada_b1 = ada_pr = None
    skip = set(filter(None, args.skip.split(",")))
    def section(fn):
        """One optional section.  Single process: a failure becomes {"error": ...} on the line (never lose the primary result).
        More than one rank: the sections contain collectives, so a rank that swallowed its own exception would leave the others
        waiting in one until the RCCL timeout -- the failing rank reports on stderr, rank 0 first prints what it has, and the process
        exits non-zero so that the launcher tears the job down at once."""
        try:
            if rank == 0:
                print(f"[bench] section {fn.__name__} ...", file=sys.stderr, flush=True)
            return fn(args, dev, world, rank)
        except Exception as e:   # noqa: BLE001
            if world == 1:
                return {"error": repr(e)}
            import traceback
            traceback.print_exc()
            print(f"[bench rank {rank}] section {fn.__name__} failed: {e!r}; aborting the job (collectives inside)", file=sys.stderr, flush=True)
            if rank == 0 and primary_so_far is not None:
                print(json.dumps(dict(primary_so_far, aborted_in=fn.__name__, error=repr(e))), flush=True)
            os._exit(3)

    primary_so_far = (inp if args.workload == "inpaint" else con) if rank == 0 else None
    if not args.no_secondary:          # collective sections: every rank enters them, rank 0 gets the record
        if "occupancy" not in skip:
            occ = section(bench_occupancy)
        if "adaptive" not in skip:
            if "adaptive_b8" not in skip:
                ada = section(bench_adaptive)
            if "adaptive_b1" not in skip:
                ada_b1 = section(bench_adaptive_b1)
            if "pointrend" not in skip:
                ada_pr = section(bench_adaptive_pointrend)
    if rank == 0:
        primary, secondary = (inp, con) if args.workload == "inpaint" else (con, inp)
        out = dict(primary)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_inpaint() if args.workload == "inpaint" else cpu_baseline()
        if not args.no_secondary:
            out["occupancy"], out["adaptive_loop"], out["adaptive_loop_b1"], out["adaptive_loop_pointrend"] = occ, ada, ada_b1, ada_pr
        if secondary is not None:
            sec = dict(secondary)
            if world == 1 and not args.no_cpu_baseline:
                sec["cpu_baseline"] = cpu_baseline() if args.workload == "inpaint" else cpu_baseline_inpaint()
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
