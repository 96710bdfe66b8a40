"""Training entrypoint and loop with the reference's flags, step order and method contract (CLIPSelf, RegionCLIP)."""
