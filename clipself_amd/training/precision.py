"""`--precision` handling.  The reference picks a torch autocast context (src/training/precision.py:5-12); the HIP
engine always computes with bf16 MFMA operands + fp32 accumulation/statistics/master weights, so every mode maps
to a null context and no GradScaler is needed (bf16 has fp32's exponent range)."""
from contextlib import suppress


def get_autocast(precision):
    return suppress
