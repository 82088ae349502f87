"""Diffusion half of the hot path: SD-1.5-inpainting UNet / VAE / DDIM on hand-written gfx950 kernels."""
