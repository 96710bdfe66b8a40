"""CPU: the C-ABI shared library loads and exports every symbol declared in include/clipself_hip.h
(no compute calls -- there is no GPU in the build container)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "clipself_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from clipself_amd import hip
    if not hip.library_path().exists():
        hip.build_library()
    lib = hip.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/clipself_hip.h but not exported"
    assert sorted(hip.SIGNATURES) == names, "hip.SIGNATURES and the header disagree"


def test_ops_refuse_to_run_without_gpu():
    import torch
    from clipself_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip.HipOps()


def test_every_entry_point_has_a_tensor_level_wrapper_and_a_reference_op():
    """HipOps (product) and RefOps (test infrastructure) both cover the whole C ABI: a wrapper silently dropped from either would
    otherwise only surface on the GPU box."""
    from clipself_amd import hip
    from oracle.ops_ref import RefOps
    renamed = {"cs_crop_resize_u8": "crop_resize", "cs_resize_bilinear_f32": "resize_bilinear"}
    for sym in hip.SIGNATURES:
        if sym in ("cs_last_error", "cs_crop_resize_workspace"):
            continue
        if sym in ("cs_num_compute_units", "cs_stream_create_cu_mask", "cs_stream_destroy"):       # runtime plumbing: nothing to restate
            assert callable(getattr(hip.HipOps, sym[len("cs_"):], None)), f"HipOps.{sym[3:]} missing"
            continue
        name = renamed.get(sym, sym[len("cs_"):])
        assert callable(getattr(hip.HipOps, name, None)), f"HipOps.{name} missing for {sym}"
        assert callable(getattr(RefOps, name, None)), f"RefOps.{name} missing for {sym}"


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/clipself_hip.h is the binding contract for non-C++ hosts: it must compile as C99 on its own, and a C program that takes the
    address of every declared entry point must link against libclipself_hip.so (no call is made -- there is no GPU here)."""
    import shutil
    import subprocess
    from clipself_amd import hip
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    if not hip.library_path().exists():
        hip.build_library()
    names = _declared()
    src = tmp_path / "bind.c"
    src.write_text('#include "clipself_hip.h"\n#include <stdio.h>\nint main(void) {\n    const void* table[] = {\n'
                   + "".join(f"        (const void*){n},\n" for n in names)
                   + '    };\n    printf("%d\\n", (int)(sizeof table / sizeof table[0]));\n    return 0;\n}\n')
    lib = hip.library_path()
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(tmp_path / "bind"),
                        f"-L{lib.parent}", "-l:" + lib.name, f"-Wl,-rpath,{lib.parent}", "-Wl,--unresolved-symbols=ignore-in-shared-libs"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_compute_unit_reservations_compose():
    """HipOps folds three requests into cs_gemm_nt's flags bits 20-27 (compute units a persistent grid leaves free): RCCL's reserve
    (training/distributed.py), the share a prefetched teacher pass leaves to the student, and the cap of the student's own grids.  Teacher side:
    reserve + share; student side: max(reserve, CUs - cap); never fewer than 8 workgroups.  (No GPU: the object is built around the logic.)"""
    from clipself_amd import hip
    ops = hip.HipOps.__new__(hip.HipOps)
    ops.gemm_flags, ops._rccl_reserve, ops._share, ops._cap, ops.num_cus = 0x90, 0, 0, 0, 256
    ops.reserve_compute_units(16)
    assert ops.persistent_grid() == 240 and ops.gemm_flags & 0xFFFFF == 0x90
    ops.share_compute_units(48)                      # the teacher inside the RCCL window of a partitioned run
    assert ops.persistent_grid() == 256 - 64
    ops.reserve_compute_units(0)
    assert ops.persistent_grid() == 208
    ops.share_compute_units(0)
    assert ops.persistent_grid() == 256 and ops.gemm_flags == 0x90
    ops.cap_compute_units(48)                        # the student beside a prefetched pass ...
    assert ops.persistent_grid() == 48
    ops.reserve_compute_units(16)                    # ... with gradient buckets in flight: the cap already leaves RCCL its CUs
    assert ops.persistent_grid() == 48
    ops.cap_compute_units(0)
    assert ops.persistent_grid() == 240
    ops.reserve_compute_units(0)
    with pytest.raises(ValueError):
        ops.cap_compute_units(4)                     # fewer than 8 workgroups: cs_persistent_cap() would ignore it
    ops._cap = 0
    ops.num_cus = 304                                # a 304-CU part (MI300X): the flags field has 8 bits
    ops.cap_compute_units(56)                        # would leave 248 CUs free: encodable
    assert ops.persistent_grid() == 56
    with pytest.raises(ValueError):
        ops.cap_compute_units(48)                    # would leave 256: not encodable -- a ValueError at the call, not an assert inside a step


def test_tower_partition_arguments_are_validated_at_construction(monkeypatch):
    """bad CLIPSELF_PARTITION_* values surface when the method is built (or when the device's CU count is first known), as ValueError."""
    from types import SimpleNamespace
    from clipself_amd.training.clipself import CLIPSelf
    CLIPSelf()                                       # off by default
    CLIPSelf(partition_cus=48)
    for kw in (dict(partition_cus=4), dict(partition_cus=-8), dict(partition_cus=44, partition_mask=True), dict(partition_cus=48, partition_cap=4)):
        with pytest.raises(ValueError):
            CLIPSelf(**kw)
    monkeypatch.setenv("CLIPSELF_PARTITION_CUS", "5")
    with pytest.raises(ValueError):
        CLIPSelf()
    monkeypatch.delenv("CLIPSELF_PARTITION_CUS")
    m = CLIPSelf(partition_cus=48)
    with pytest.raises(ValueError):
        m._check_partition(SimpleNamespace(num_compute_units=lambda: 304))     # the student's cap of 48 would leave 256 CUs free
    m = CLIPSelf(partition_cus=48, partition_cap=64)
    m._check_partition(SimpleNamespace(num_compute_units=lambda: 304))
    assert m._partition_checked


def test_rotary_table_layout_is_checked_by_the_wrapper():
    """Round 6: the attention kernels read the rotary tables separably and (attn_fwd4_kernel) once per frequency -- the layout rope.py:118-142 builds
    is a documented precondition of the C ABI.  HipOps verifies it once per table tensor: the oracle's tables and identity tables (the OpenAI-CLIP
    family) pass, also as inference tensors; a table whose row part differs from its column part, whose pair entries differ, or that is not
    separable raises instead of letting a kernel return wrong numbers.  (No GPU: the object is built around the logic.)"""
    import torch
    from clipself_amd import hip
    from oracle.eva_ref import rope_tables
    ops = hip.HipOps.__new__(hip.HipOps)
    cos, sin = rope_tables(14, 64)
    ops._check_rope_tables(cos, sin, 197)
    ops._check_rope_tables(cos, sin, 197)                          # cached
    ops._check_rope_tables(torch.ones(196, 64), torch.zeros(196, 64), 197)
    with torch.inference_mode():
        c2, s2 = cos.clone(), sin.clone()
    ops._check_rope_tables(c2, s2, 197)
    for poke in ((5, 3), (5, 40), (17, 2)):                          # pair partner / column part of one token / row part of one token
        bad = cos.clone()
        bad[poke] += 0.25
        with pytest.raises(ValueError):
            ops._check_rope_tables(bad, sin, 197)
    with pytest.raises(ValueError):
        ops._check_rope_tables(cos[:195], sin[:195], 196)          # not a square grid
    c3, s3 = rope_tables(14, 64)
    c3 = c3.view(14, 14, 64).clone()
    c3[:, :, 32:] = c3[:, :, 32:].flip(1)                          # separable, pairs intact, but column part != row part
    with pytest.raises(ValueError):
        ops._check_rope_tables(c3.view(196, 64), s3, 197)
