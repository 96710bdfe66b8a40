"""clipself_amd -- the CLIPSelf distillation step on MI355X (gfx950): hand-written HIP kernels behind a C ABI (csrc/, hip.py), the
step engine (engine.py) and the reference-compatible Python surface (open_clip/, training/)."""
