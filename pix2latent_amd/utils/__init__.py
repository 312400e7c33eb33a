"""Host-side helpers of the hot path: `function_hooks` (per-candidate latent hooks applied on the
contiguous buffer), `image` / `video` (PIL read, grids, gif / mp4 behind soft dependencies), `misc`
(progress printing, HWC/CHW conversions), `checkpoint` (upstream BigGAN / LPIPS / StyleGAN2 files ->
the flat weight dicts of the HIP models), `synthetic` (seeded weights and targets for tests and
bench.py)."""
