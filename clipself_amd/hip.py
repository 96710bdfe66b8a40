"""ctypes binding of the C-ABI kernel library (include/clipself_hip.h) + a thin tensor-level wrapper.

PyTorch is used here only as plumbing: device allocation (`torch.empty(device='cuda')`), the current HIP stream
and `torch.distributed`.  Every op below is one call into libclipself_hip.so with raw device pointers.

There is NO fallback: if the library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import torch

_CSRC = Path(__file__).resolve().parent / "csrc"
_LIB_PATH = Path(os.environ["CLIPSELF_HIP_LIB"]) if os.environ.get("CLIPSELF_HIP_LIB") else _CSRC / "libclipself_hip.so"   # override: A/B of two builds

EPI_BF16, EPI_F32, EPI_RESID_F32, EPI_SWIGLU_BF16, EPI_ATOMIC_F32, EPI_PATCH_F32, EPI_RESID_LN_F32, EPI_GELU_BF16, EPI_QGELU_BF16 = range(9)
DX_BF16, DX_F32_ASSIGN, DX_F32_ACCUM = range(3)

_vp, _i, _l, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/clipself_hip.h
SIGNATURES = {
    "cs_last_error": (ctypes.c_char_p, []),
    "cs_gemm_nt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_crop_resize_workspace": (_sz, [_i, _i, _i]),
    "cs_crop_resize_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cs_quant_rows_fp8": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _vp]),
    "cs_gemm_nt_f8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_resize_bilinear_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cs_gemm_wgrad_workspace": (_sz, [_i, _i, _i]),
    "cs_gemm_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_gemm_wgrad_tn_workspace": (_sz, [_i, _i, _i]),
    "cs_gemm_wgrad_tn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_gemm_nt_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_gemm_nt_ln_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_ln_stats_finalize": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "cs_attn_fwd_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "cs_layernorm_fwd": (_i, [_vp, _i, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _f, _vp]),
    "cs_layernorm_fwd_q8": (_i, [_vp, _i, _l, _vp, _vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _i, _i, _f, _vp]),
    "cs_layernorm_fwd_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _f, _vp]),
    "cs_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "cs_layernorm_bwd": (_i, [_vp, _l, _vp, _i, _l, _vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _vp, _vp, _l, _vp, _i, _i, _vp]),
    "cs_layernorm_bwd_q8": (_i, [_vp, _l, _vp, _i, _l, _vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _vp, _vp, _l, _vp, _vp, _l, _vp, _i, _i, _vp]),
    "cs_l2norm_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "cs_l2norm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cs_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "cs_attn_cls_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "cs_attn_query_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "cs_attn_bwd_workspace": (_sz, [_i, _i, _i]),
    "cs_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "cs_swiglu_fwd": (_i, [_vp, _l, _vp, _l, _i, _i, _vp]),
    "cs_swiglu_bwd": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp]),
    "cs_swiglu_bwd_colsum": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _vp]),
    "cs_swiglu_bwd_q8": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _vp]),
    "cs_gelu_fwd": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp]),
    "cs_gelu_bwd": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _vp]),
    "cs_cast_f32_bf16": (_i, [_vp, _vp, _l, _vp]),
    "cs_transpose_bf16": (_i, [_vp, _l, _vp, _l, _i, _i, _vp]),
    "cs_transpose_bf16_batched": (_i, [_vp, _i, _i, _vp]),
    "cs_colsum_workspace": (_sz, [_i, _i]),
    "cs_colsum_bf16": (_i, [_vp, _l, _vp, _vp, _i, _i, _vp]),
    "cs_im2row": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "cs_cls_row": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "cs_roialign_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_roialign_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cs_cosine_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "cs_cosine_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "cs_fed_bce_fwd": (_i, [_vp, _l, _vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "cs_fed_bce_bwd": (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _f, _f, _vp, _vp]),
    "cs_num_compute_units": (_i, []),
    "cs_stream_create_cu_mask": (_i, [_i, _i, ctypes.POINTER(_vp)]),
    "cs_stream_destroy": (_i, [_vp]),
    "cs_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp]),
}


def library_path() -> Path:
    return _LIB_PATH


def build_library(force: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        for o in _CSRC.glob("_obj_*.o"):
            o.unlink()
    subprocess.run(["bash", str(_CSRC / "build.sh")], check=True)
    return _LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with clipself_amd/csrc/build.sh (or __graft_entry__.build()). "
                "There is no CPU fallback for the CLIPSelf hot path.")
        lib = ctypes.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _dt(t) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError(f"unsupported dtype {t.dtype}")


class HipOps:
    """Tensor-level face of the C ABI.  Tensors are 2-D (rows, cols) views whose last stride is 1; row strides
    become the `ld*` arguments.  Outputs are written in place into caller-allocated tensors."""

    name = "hip"

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a ROCm device (torch.cuda.is_available() is False); no CPU fallback exists")
        # OR-ed into the flags of every GEMM call: bits 20-27 = compute units the persistent GEMM kernels leave free.  Two callers
        # withhold CUs: training/distributed.py (a few for RCCL's reduction kernels while gradient buckets are in flight) and the
        # single-GPU tower partition of training/clipself.py (the frozen teacher's prefetched pass leaves `share` CUs to the student,
        # whose own persistent GEMMs are capped at `cap` workgroups meanwhile).
        self.gemm_flags = 0
        self._rccl_reserve = 0          # reserve_compute_units()
        self._share = 0                 # share_compute_units(): CUs given to the other tower, added to the RCCL reserve
        self._cap = 0                   # cap_compute_units(): upper bound of a persistent grid (0 = none)
        self.num_cus = int(self.lib.cs_num_compute_units())

    def _update_flags(self):
        n = self._rccl_reserve + self._share
        if self._cap:
            n = max(n, self.num_cus - self._cap)
        # a ValueError, not an assert (python -O), and at the call that asked for it: the field of the GEMM flags has 8 bits, and a
        # persistent grid keeps at least 8 workgroups (one per XCD)
        if not (0 <= n <= self.num_cus - 8 and n < 256):
            raise ValueError(f"persistent GEMM grids keep at least 8 workgroups and can leave at most 255 compute units free "
                             f"(asked to leave {n} of {self.num_cus} CUs: RCCL reserve {self._rccl_reserve}, tower share {self._share}, cap {self._cap})")
        self.gemm_flags = (self.gemm_flags & ~(255 << 20)) | (int(n) << 20)

    def reserve_compute_units(self, n: int):
        """Persistent GEMM grids leave n compute units free from now on (0 = none): room for RCCL's kernels in data-parallel runs."""
        self._rccl_reserve = int(n)
        self._update_flags()

    def share_compute_units(self, n: int):
        """... and n more for the other tower of the step (the frozen teacher's side of the single-GPU partition)."""
        self._share = int(n)
        self._update_flags()

    def cap_compute_units(self, n: int):
        """Persistent GEMM grids use at most n workgroups (0 = no cap): the student's side of the partition."""
        self._cap = int(n)
        self._update_flags()

    def persistent_grid(self) -> int:
        """Workgroups a persistent GEMM launches under the current reservation (diagnostics, tests)."""
        return self.num_cus - ((self.gemm_flags >> 20) & 255)

    def num_compute_units(self) -> int:
        return self.num_cus

    def stream_create_cu_mask(self, first_cu: int, n_cus: int):
        """torch stream whose kernels run on compute units [first_cu, first_cu + n_cus) only (cs_stream_create_cu_mask)."""
        h = ctypes.c_void_p()
        self._ok(self.lib.cs_stream_create_cu_mask(int(first_cu), int(n_cus), ctypes.byref(h)), "cs_stream_create_cu_mask")
        s = torch.cuda.ExternalStream(h.value)
        s._cs_handle = h.value
        return s

    def stream_destroy(self, stream):
        torch.cuda.synchronize()
        self._ok(self.lib.cs_stream_destroy(ctypes.c_void_p(stream._cs_handle)), "cs_stream_destroy")

    # -- helpers ---------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def _ok(self, rc, who):
        if rc != 0:
            raise RuntimeError(f"{who} failed ({rc}): {self.lib.cs_last_error().decode()}")

    @staticmethod
    def _chk(*ts):
        for t in ts:
            if t is not None and not t.is_cuda:
                raise RuntimeError("HipOps received a non-GPU tensor; the hot path has no CPU fallback")

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device="cuda")

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device="cuda")

    # -- ops -------------------------------------------------------------------------------------
    def gemm_nt(self, A, B, C, bias=None, extra=None, epi=EPI_BF16, splits=1, group=0, flags=0):
        self._chk(A, B, C, bias, extra)
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(-1) == 1
        if extra is not None and epi in (EPI_RESID_F32, EPI_PATCH_F32):
            assert extra.stride(0) == C.stride(0), "extra must share C's row stride"
        self._ok(self.lib.cs_gemm_nt(_p(A), _p(B), _p(C), _p(bias), _p(extra), M, N, K, A.stride(0), B.stride(0),
                                     C.stride(0), epi, splits, group, flags | self.gemm_flags, self._stream()), "cs_gemm_nt")

    def gemm_nt_ln(self, A, B, C, bias=None, extra=None, ln_mean=None, ln_rstd=None, ln_colsum=None, stats_part=None, xb_out=None,
                   epi=EPI_RESID_LN_F32, group=0, flags=0):
        self._chk(A, B, C, bias, extra, ln_mean, ln_rstd, ln_colsum, stats_part, xb_out)
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(-1) == 1
        if extra is not None and epi in (EPI_RESID_F32, EPI_RESID_LN_F32):
            assert extra.stride(0) == C.stride(0), "extra must share C's row stride"
        if stats_part is not None:
            slices = 4 * ((group + 127) // 128) if epi == EPI_SWIGLU_BF16 else (N + 63) // 64
            assert stats_part.is_contiguous() and stats_part.shape[0] >= slices and stats_part.shape[1] == M
        if xb_out is not None:
            assert xb_out.dtype == torch.bfloat16 and xb_out.stride(1) == 1 and xb_out.shape[0] == M
        self._ok(self.lib.cs_gemm_nt_ln(_p(A), _p(B), _p(C), _p(bias), _p(extra), _p(ln_mean), _p(ln_rstd), _p(ln_colsum), _p(stats_part),
                                        _p(xb_out), xb_out.stride(0) if xb_out is not None else 0, M, N, K, A.stride(0), B.stride(0),
                                        C.stride(0), epi, 1, group, flags | self.gemm_flags, self._stream()), "cs_gemm_nt_ln")

    def gemm_nt_ln_split(self, A, B, hi, lo, bias, ln_mean, ln_rstd, ln_colsum, x_in=None, x_out=None, stats_part=None, flags=0):
        """Folded-LayerNorm residual GEMM on the split stream (cs_gemm_nt_ln_split): x += rstd * (A.B^T - mean * colsum) + bias with x held as
        the 16-bit planes hi (bf16 view of x, the next folded GEMM's operand) and lo (int16) -- or read from x_in / written to x_out (fp32)
        at the two ends of the tower."""
        self._chk(A, B, hi, lo, bias, ln_mean, ln_rstd, ln_colsum, x_in, x_out, stats_part)
        M, K = A.shape
        N = B.shape[0]
        assert hi.dtype == torch.bfloat16 and lo.dtype == torch.int16 and hi.shape == (M, N) and lo.shape == (M, N)
        assert hi.stride(1) == 1 and lo.stride(1) == 1 and hi.stride(0) == lo.stride(0) and hi.stride(0) % 8 == 0
        assert x_out is None or stats_part is None, "fp32 out: hi / lo are only read, no statistics epilogue"
        x = x_in if x_in is not None else x_out
        assert x is None or (x.dtype == torch.float32 and x.shape == (M, N) and x.stride(1) == 1)
        self._ok(self.lib.cs_gemm_nt_ln_split(_p(A), _p(B), _p(bias), _p(ln_mean), _p(ln_rstd), _p(ln_colsum), _p(x_in), _p(x_out), _p(hi), _p(lo),
                                              hi.stride(0), _p(stats_part), M, N, K, A.stride(0), B.stride(0), x.stride(0) if x is not None else N,
                                              flags | self.gemm_flags, self._stream()), "cs_gemm_nt_ln_split")

    def quant_rows_fp8(self, x, q, scale):
        """bf16 [M,K] -> e4m3 bytes q [M,Kp] (uint8 / float8 storage, Kp = K rounded up to 128, padding zeroed) + fp32 row scales [M]."""
        self._chk(x, q, scale)
        M, K = x.shape
        assert x.dtype == torch.bfloat16 and x.stride(1) == 1 and q.element_size() == 1 and q.stride(1) == 1 and q.shape[1] == (K + 127) // 128 * 128
        self._ok(self.lib.cs_quant_rows_fp8(_p(x), x.stride(0), _p(q), q.stride(0), _p(scale), M, K, self._stream()), "cs_quant_rows_fp8")

    def gemm_nt_f8(self, A8, B8, C, row_scale, col_scale, bias=None, extra=None, epi=EPI_BF16, flags=0):
        """C = row_scale[m] * col_scale[n] * (A8 . B8^T) + bias (+ extra): e4m3 operands [M,Kp] / [N,Kp] from quant_rows_fp8."""
        self._chk(A8, B8, C, row_scale, col_scale, bias, extra)
        M, K8 = A8.shape
        N = B8.shape[0]
        assert B8.shape[1] == K8 and A8.element_size() == 1 and B8.element_size() == 1 and C.stride(-1) == 1
        if extra is not None:
            assert extra.stride(0) == C.stride(0), "extra must share C's row stride"
        self._ok(self.lib.cs_gemm_nt_f8(_p(A8), _p(B8), _p(C), _p(bias), _p(extra), _p(row_scale), _p(col_scale), M, N, K8, A8.stride(0), B8.stride(0),
                                        C.stride(0), epi, flags | self.gemm_flags, self._stream()), "cs_gemm_nt_f8")

    def crop_resize(self, image_u8, boxes, size, pad_center=True, mean=(0.48145466, 0.4578275, 0.40821073),
                    std=(0.26862954, 0.26130258, 0.27577711), out=None):
        """image_u8 [H,W,3] uint8 on the GPU, boxes [K,4] f32 pixel xyxy -> [K,3,size,size] f32: crop, Pillow-exact bicubic resize of
        the longest side to `size`, zero pad (centred or right/bottom), /255, normalise (defaults: the OpenAI CLIP statistics)."""
        self._chk(image_u8, boxes, out)
        assert image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3 and image_u8.is_contiguous()
        boxes = boxes.to(torch.float32).contiguous()
        H, W, K = image_u8.shape[0], image_u8.shape[1], boxes.shape[0]
        if out is None:
            out = torch.empty((K, 3, size, size), dtype=torch.float32, device=image_u8.device)
        need = int(self.lib.cs_crop_resize_workspace(H, K, size))
        ws = getattr(self, "_crop_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._crop_ws = torch.empty(need, dtype=torch.uint8, device=image_u8.device)
        m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        self._ok(self.lib.cs_crop_resize_u8(_p(image_u8), H, W, _p(boxes), K, size, int(bool(pad_center)), m3, s3, _p(out), _p(ws),
                                            self._stream()), "cs_crop_resize_u8")
        return out

    def resize_bilinear(self, x, size):
        """x fp32 [B,C,H,W] -> [B,C,size,size], torch's bilinear / align_corners=False arithmetic (the --multiscale resize)."""
        self._chk(x)
        assert x.dtype == torch.float32 and x.dim() == 4
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, C, size, size), dtype=torch.float32, device=x.device)
        self._ok(self.lib.cs_resize_bilinear_f32(_p(x), _p(out), B * C, H, W, size, size, self._stream()), "cs_resize_bilinear_f32")
        return out

    def gemm_wgrad_workspace(self, M, N, K) -> int:
        return int(self.lib.cs_gemm_wgrad_workspace(M, N, K))

    def gemm_wgrad(self, A, B, dW, workspace):
        """dW[M,N] += A[M,K] . B[N,K]^T (split-K through `workspace`, a uint8/any buffer of >= gemm_wgrad_workspace bytes)."""
        self._chk(A, B, dW, workspace)
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K and A.stride(1) == 1 and B.stride(1) == 1 and dW.stride(1) == 1 and dW.dtype == torch.float32
        assert workspace.numel() * workspace.element_size() >= self.gemm_wgrad_workspace(M, N, K)
        self._ok(self.lib.cs_gemm_wgrad(_p(A), _p(B), _p(dW), _p(workspace), M, N, K, A.stride(0), B.stride(0), dW.stride(0),
                                        self._stream()), "cs_gemm_wgrad")

    def gemm_wgrad_tn_workspace(self, N, K, tokens) -> int:
        """Bytes of workspace for gemm_wgrad_tn, 0 when the shape is outside its coverage (take the transposing path then)."""
        return int(self.lib.cs_gemm_wgrad_tn_workspace(N, K, tokens))

    def gemm_wgrad_tn(self, dY, X, dW, workspace):
        """dW[N,K] += dY[tokens,N]^T . X[tokens,K] from the token-major operands (no transposed copies)."""
        self._chk(dY, X, dW, workspace)
        T, N = dY.shape
        K = X.shape[1]
        assert X.shape[0] == T and dY.stride(1) == 1 and X.stride(1) == 1 and dW.stride(1) == 1 and dW.dtype == torch.float32
        assert workspace.numel() * workspace.element_size() >= self.gemm_wgrad_tn_workspace(N, K, T) > 0
        rc = self.lib.cs_gemm_wgrad_tn(_p(dY), _p(X), _p(dW), _p(workspace), N, K, T, dY.stride(0), X.stride(0), dW.stride(0), self._stream())
        if rc == 1:
            raise RuntimeError("cs_gemm_wgrad_tn: shape outside the kernel's coverage (check gemm_wgrad_tn_workspace first)")
        self._ok(rc, "cs_gemm_wgrad_tn")

    def ln_stats_finalize(self, part, npp, C, mean, rstd, eps=1e-6):
        self._chk(part, mean, rstd)
        P, M = part.shape[0], part.shape[1]
        assert part.is_contiguous() and part.shape[2] == 2
        self._ok(self.lib.cs_ln_stats_finalize(_p(part), P, npp, C, M, eps, _p(mean), _p(rstd), self._stream()), "cs_ln_stats_finalize")

    def _check_rope_tables(self, cos, sin, Ntok):
        """The forward kernels read the rotary tables separably and (round 6, attn_fwd4_kernel) ONCE per frequency: for a g x g grid, dims
        [0, 32) of token (r, c) depend on r only, dims [32, 64) on c only, the row part of grid row i equals the column part of grid column i,
        and the two dims of a rotation pair share their entry -- exactly what rope.py:118-142 builds (one `freqs` tensor, repeated for the
        pair, broadcast over rows and columns).  The C ABI documents this as a precondition (include/clipself_hip.h); this wrapper checks it
        once per table tensor (a few reductions and one host read-back) and raises instead of letting a kernel return wrong numbers."""
        def ver(t):
            try:
                return t._version
            except RuntimeError:               # inference tensors carry no version counter
                return -1
        key = (cos.data_ptr(), sin.data_ptr(), ver(cos), ver(sin), Ntok)
        seen = self.__dict__.setdefault("_rope_ok", set())
        if key in seen:
            return
        g = int(round((Ntok - 1) ** 0.5))
        if g * g != Ntok - 1 or tuple(cos.shape) != (Ntok - 1, 64) or tuple(sin.shape) != (Ntok - 1, 64):
            raise ValueError(f"rotary tables must be [g*g, 64] for a square token grid, got {tuple(cos.shape)} for {Ntok} tokens")
        for t in (cos, sin):
            v = t.view(g, g, 64)
            ok = (torch.equal(v[:, :, :32], v[:, :1, :32].expand(g, g, 32)) and torch.equal(v[:, :, 32:], v[:1, :, 32:].expand(g, g, 32))
                  and torch.equal(v[:, 0, :32], v[0, :, 32:]) and torch.equal(v[..., 0::2], v[..., 1::2]))
            if not ok:
                raise ValueError("rotary tables are not in the layout of rope.py:118-142 (separable, row part == column part, one entry per "
                                 "rotation pair): the attention kernels cannot read them")
        if len(seen) > 64:
            seen.clear()
        seen.add(key)

    def attn_fwd_stats(self, qkv, cos, sin, out, lse, stats_part, B, Ntok, H, scale):
        self._chk(qkv, cos, sin, out, lse, stats_part)
        self._check_rope_tables(cos, sin, Ntok)
        assert stats_part.is_contiguous() and tuple(stats_part.shape) == (H, B * Ntok, 2)
        self._ok(self.lib.cs_attn_fwd_stats(_p(qkv), _p(cos), _p(sin), _p(out), _p(lse), _p(stats_part), B, Ntok, H, qkv.stride(0),
                                            out.stride(0), scale, self._stream()), "cs_attn_fwd_stats")

    def layernorm_fwd(self, x, gamma, beta, y, mean=None, rstd=None, eps=1e-6, q8=None, q_scale=None):
        if q8 is not None:
            return self.layernorm_fwd_q8(x, gamma, beta, y, q8, q_scale, mean, rstd, eps)
        self._chk(x, gamma, beta, y, mean, rstd)
        M, C = x.shape
        self._ok(self.lib.cs_layernorm_fwd(_p(x), _dt(x), x.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0) if y is not None else 0,
                                           _p(mean), _p(rstd), M, C, eps, self._stream()), "cs_layernorm_fwd")

    def layernorm_fwd_q8(self, x, gamma, beta, y, q8, q_scale, mean=None, rstd=None, eps=1e-6):
        """LayerNorm forward that also writes the e4m3 copy of y for the fp8 GEMM: q8 [M, Kp] (1-byte elements, Kp >= C rounded up to 128,
        padding zero) + q_scale [M]; bit-identical to quant_rows_fp8(y)."""
        self._chk(x, gamma, beta, y, mean, rstd, q8, q_scale)
        M, C = x.shape
        assert q8.element_size() == 1 and q8.stride(1) == 1 and q8.shape[1] >= (C + 127) // 128 * 128 and q_scale is not None and y is not None
        self._ok(self.lib.cs_layernorm_fwd_q8(_p(x), _dt(x), x.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), _p(mean), _p(rstd),
                                              _p(q8), q8.stride(0), _p(q_scale), M, C, eps, self._stream()), "cs_layernorm_fwd_q8")

    def layernorm_fwd_f32(self, x, gamma, beta, y, mean=None, rstd=None, eps=1e-5):
        """fp32 rows -> fp32 rows (ln_pre of the OpenAI-CLIP ViT: its output is the residual stream)."""
        self._chk(x, gamma, beta, y, mean, rstd)
        M, C = x.shape
        assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.stride(1) == 1 and y.stride(1) == 1
        assert x.data_ptr() != y.data_ptr(), "layernorm_fwd_f32 is out of place"
        self._ok(self.lib.cs_layernorm_fwd_f32(_p(x), x.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), _p(mean), _p(rstd), M, C, eps,
                                               self._stream()), "cs_layernorm_fwd_f32")

    def layernorm_bwd_workspace(self, M, C) -> int:
        return int(self.lib.cs_layernorm_bwd_workspace(M, C))

    def layernorm_bwd_q8(self, dy, x, gamma, mean, rstd, dx, dx_mode, dgamma, dbeta, accumulate, workspace, dx_copy, copy_colsum, q8, q_scale):
        """layernorm_bwd whose bf16 copy also leaves as e4m3 bytes q8 [M, >= C rounded up to 128] + fp32 row scales (= quant_rows_fp8(dx_copy))."""
        self._chk(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, dx_copy, copy_colsum, q8, q_scale)
        M, C = x.shape
        assert dx_copy is not None and dx_copy.dtype == torch.bfloat16 and dx_copy.shape == (M, C) and dx_copy.stride(1) == 1
        assert q8.element_size() == 1 and q8.stride(1) == 1 and q8.shape[1] >= (C + 127) // 128 * 128 and q_scale is not None
        self._ok(self.lib.cs_layernorm_bwd_q8(_p(dy), dy.stride(0), _p(x), _dt(x), x.stride(0), _p(gamma), _p(mean), _p(rstd),
                                              _p(dx), dx_mode, dx.stride(0), _p(dgamma), _p(dbeta), int(accumulate),
                                              _p(workspace), _p(dx_copy), dx_copy.stride(0), _p(copy_colsum), _p(q8), q8.stride(0), _p(q_scale),
                                              M, C, self._stream()), "cs_layernorm_bwd_q8")

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dx_mode, dgamma=None, dbeta=None, accumulate=False, workspace=None,
                      dx_copy=None, copy_colsum=None, q8=None, q_scale=None):
        """dx_copy (fp32 dx modes): bf16 copy of the updated dx rows; copy_colsum [C]: (+)= its column sums (a bias gradient);
        q8 / q_scale: the e4m3 copy of dx_copy as well (layernorm_bwd_q8)."""
        if q8 is not None:
            return self.layernorm_bwd_q8(dy, x, gamma, mean, rstd, dx, dx_mode, dgamma, dbeta, accumulate, workspace, dx_copy, copy_colsum,
                                         q8, q_scale)
        self._chk(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, dx_copy, copy_colsum)
        M, C = x.shape
        if dx_copy is not None:
            assert dx_copy.dtype == torch.bfloat16 and dx_copy.shape == (M, C) and dx_copy.stride(1) == 1
        self._ok(self.lib.cs_layernorm_bwd(_p(dy), dy.stride(0), _p(x), _dt(x), x.stride(0), _p(gamma), _p(mean), _p(rstd),
                                           _p(dx), dx_mode, dx.stride(0), _p(dgamma), _p(dbeta), int(accumulate),
                                           _p(workspace), _p(dx_copy), dx_copy.stride(0) if dx_copy is not None else 0, _p(copy_colsum),
                                           M, C, self._stream()), "cs_layernorm_bwd")

    def l2norm_fwd(self, x, y, inv_norm, eps=1e-12):
        self._chk(x, y, inv_norm)
        M, C = x.shape
        assert x.is_contiguous() and y.is_contiguous()
        self._ok(self.lib.cs_l2norm_fwd(_p(x), _p(y), _p(inv_norm), M, C, eps, self._stream()), "cs_l2norm_fwd")

    def l2norm_bwd(self, dy, y, inv_norm, dx):
        self._chk(dy, y, inv_norm, dx)
        M, C = y.shape
        assert dy.is_contiguous() and y.is_contiguous() and dx.is_contiguous()
        self._ok(self.lib.cs_l2norm_bwd(_p(dy), _p(y), _p(inv_norm), _p(dx), M, C, self._stream()), "cs_l2norm_bwd")

    def attn_fwd(self, qkv, cos, sin, out, lse, B, Ntok, H, scale):
        self._chk(qkv, cos, sin, out, lse)
        self._check_rope_tables(cos, sin, Ntok)
        self._ok(self.lib.cs_attn_fwd(_p(qkv), _p(cos), _p(sin), _p(out), _p(lse), B, Ntok, H, qkv.stride(0), out.stride(0),
                                      scale, self._stream()), "cs_attn_fwd")

    def attn_cls_fwd(self, q, kv, cos, sin, out, B, Ntok, H, scale):
        self._chk(q, kv, cos, sin, out)
        self._ok(self.lib.cs_attn_cls_fwd(_p(q), _p(kv), _p(cos), _p(sin), _p(out), B, Ntok, H, q.stride(0), kv.stride(0),
                                          out.stride(0), scale, self._stream()), "cs_attn_cls_fwd")

    def attn_query_fwd(self, q, kv, allow, out, B, Q, Ntok, H, scale):
        """Q extra query rows per image against the image's keys / values; allow [B*Q, Ntok] uint8 (1 = may attend).  Inference only."""
        self._chk(q, kv, allow, out)
        assert allow.dtype == torch.uint8 and allow.is_contiguous() and tuple(allow.shape) == (B * Q, Ntok)
        self._ok(self.lib.cs_attn_query_fwd(_p(q), _p(kv), _p(allow), _p(out), B, Q, Ntok, H, q.stride(0), kv.stride(0), out.stride(0),
                                            scale, self._stream()), "cs_attn_query_fwd")

    def attn_bwd_workspace(self, B, Ntok, H) -> int:
        return int(self.lib.cs_attn_bwd_workspace(B, Ntok, H))

    def attn_bwd(self, qkv, o, dout, lse, cos, sin, dqkv, workspace, B, Ntok, H, scale):
        self._chk(qkv, o, dout, lse, cos, sin, dqkv, workspace)
        assert o.stride(0) == dout.stride(0) and qkv.stride(0) == dqkv.stride(0)
        self._ok(self.lib.cs_attn_bwd(_p(qkv), _p(o), _p(dout), _p(lse), _p(cos), _p(sin), _p(dqkv), _p(workspace), B, Ntok, H,
                                      qkv.stride(0), o.stride(0), scale, self._stream()), "cs_attn_bwd")

    def swiglu_fwd(self, x12, h):
        self._chk(x12, h)
        M, Hd = h.shape
        self._ok(self.lib.cs_swiglu_fwd(_p(x12), x12.stride(0), _p(h), h.stride(0), M, Hd, self._stream()), "cs_swiglu_fwd")

    def swiglu_bwd_q8(self, dh, x12, dx12, q8, q_scale):
        """swiglu_bwd + the e4m3 copy of dx12 (bytes [M, 2*Hd rounded up to 128] + fp32 row scales) for an fp8 dgrad; = quant_rows_fp8(dx12)."""
        self._chk(dh, x12, dx12, q8, q_scale)
        M, Hd = dh.shape
        assert q_scale is not None and q8.element_size() == 1 and q8.stride(1) == 1 and q8.shape[1] >= (2 * Hd + 127) // 128 * 128
        self._ok(self.lib.cs_swiglu_bwd_q8(_p(dh), dh.stride(0), _p(x12), x12.stride(0), _p(dx12), dx12.stride(0), _p(q8), q8.stride(0),
                                           _p(q_scale), M, Hd, self._stream()), "cs_swiglu_bwd_q8")

    def swiglu_bwd_colsum(self, dh, x12, dx12, colsum, workspace):
        """swiglu_bwd + colsum[2*Hd] += column sums of dx12 (the w1 | w2 bias gradients) in the same pass; = swiglu_bwd, colsum_bf16(dx12)."""
        self._chk(dh, x12, dx12, colsum, workspace)
        M, Hd = dh.shape
        assert colsum.dtype == torch.float32 and colsum.numel() >= 2 * Hd and colsum.is_contiguous()
        assert workspace.numel() * workspace.element_size() >= self.colsum_workspace(M, 2 * Hd)
        self._ok(self.lib.cs_swiglu_bwd_colsum(_p(dh), dh.stride(0), _p(x12), x12.stride(0), _p(dx12), dx12.stride(0), _p(colsum), _p(workspace),
                                               M, Hd, self._stream()), "cs_swiglu_bwd_colsum")

    def swiglu_bwd(self, dh, x12, dx12, q8=None, q_scale=None):
        if q8 is not None:
            return self.swiglu_bwd_q8(dh, x12, dx12, q8, q_scale)
        self._chk(dh, x12, dx12)
        M, Hd = dh.shape
        self._ok(self.lib.cs_swiglu_bwd(_p(dh), dh.stride(0), _p(x12), x12.stride(0), _p(dx12), dx12.stride(0), M, Hd,
                                        self._stream()), "cs_swiglu_bwd")

    def gelu_fwd(self, x, y, quick=False):
        self._chk(x, y)
        M, N = x.shape
        self._ok(self.lib.cs_gelu_fwd(_p(x), x.stride(0), _p(y), y.stride(0), M, N, int(bool(quick)), self._stream()), "cs_gelu_fwd")

    def gelu_bwd(self, dy, x, dx, quick=False):
        self._chk(dy, x, dx)
        M, N = x.shape
        self._ok(self.lib.cs_gelu_bwd(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dx), dx.stride(0), M, N, int(bool(quick)),
                                      self._stream()), "cs_gelu_bwd")

    def cast_f32_bf16(self, x, y):
        self._chk(x, y)
        assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
        self._ok(self.lib.cs_cast_f32_bf16(_p(x), _p(y), x.numel(), self._stream()), "cs_cast_f32_bf16")

    def transpose_bf16(self, inp, out):
        """out[c, r] = inp[r, c]; out is [Cc, ld_out >= R] and columns R.. are zero-filled."""
        self._chk(inp, out)
        R, Cc = inp.shape
        assert out.shape[0] == Cc and out.is_contiguous()
        self._ok(self.lib.cs_transpose_bf16(_p(inp), inp.stride(0), _p(out), out.shape[1], R, Cc, self._stream()), "cs_transpose_bf16")

    def transpose_bf16_batched(self, pairs):
        """[(inp [R, C], out [C, ld_out >= R]), ...] -> every out = inp^T (zero padded) in ONE launch.  The device descriptor table is
        cached per list of buffers (the weight shadows and their transposes keep their addresses for the life of the engine)."""
        import numpy as np
        key = tuple((a.data_ptr(), b.data_ptr(), a.shape, b.shape, a.stride(0)) for a, b in pairs)
        tables = self.__dict__.setdefault("_tr_desc_tables", {})      # a few lists at most: every trainable block / a subset after a re-lock
        cache = tables.get(key)
        if cache is None:
            if len(tables) >= 8:
                tables.clear()
            rec = np.zeros((len(pairs), 6), dtype=np.int64)
            tile0 = 0
            for i, (a, b) in enumerate(pairs):
                self._chk(a, b)
                R, Cc = a.shape
                assert b.shape[0] == Cc and b.is_contiguous() and b.shape[1] >= R and a.stride(1) == 1
                tiles_x, tiles_y = (Cc + 63) // 64, (b.shape[1] + 63) // 64
                rec[i] = (a.data_ptr(), b.data_ptr(), a.stride(0), b.shape[1], R | (Cc << 32), tile0 | (tiles_x << 32))
                tile0 += tiles_x * tiles_y
            cache = tables[key] = (key, torch.from_numpy(rec).cuda(), tile0)
        _, desc, total = cache
        self._ok(self.lib.cs_transpose_bf16_batched(_p(desc), len(pairs), total, self._stream()), "cs_transpose_bf16_batched")

    def colsum_workspace(self, M, N) -> int:
        return int(self.lib.cs_colsum_workspace(M, N))

    def colsum_bf16(self, x, out, workspace=None):
        """out[n] += sum_m x[m, n] in a fixed order (row-block partials through `workspace`, >= colsum_workspace(M, N) bytes; allocated
        here when not given)."""
        self._chk(x, out, workspace)
        M, N = x.shape
        need = self.colsum_workspace(M, N)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        self._ok(self.lib.cs_colsum_bf16(_p(x), x.stride(0), _p(out), _p(workspace), M, N, self._stream()), "cs_colsum_bf16")

    def im2row(self, img, out, p):
        self._chk(img, out)
        B, _, S, _ = img.shape
        assert img.is_contiguous()
        self._ok(self.lib.cs_im2row(_p(img), _dt(img), _p(out), B, S, p, out.stride(0), self._stream()), "cs_im2row")

    def cls_row(self, x, cls, pos):
        self._chk(x, cls, pos)
        B, Ntok, C = x.shape
        self._ok(self.lib.cs_cls_row(_p(x), _p(cls), _p(pos), B, Ntok, C, self._stream()), "cs_cls_row")

    def roialign_fwd(self, feat, rois, pooled, grid_h, grid_w, tok_off):
        self._chk(feat, rois, pooled)
        B, Ntok, E = feat.shape
        self._ok(self.lib.cs_roialign_fwd(_p(feat), _p(rois), _p(pooled), rois.shape[0], Ntok, grid_h, grid_w, E, tok_off,
                                          self._stream()), "cs_roialign_fwd")

    def roialign_bwd(self, dpooled, rois, dfeat, grid_h, grid_w, tok_off):
        self._chk(dpooled, rois, dfeat)
        B, Ntok, E = dfeat.shape
        self._ok(self.lib.cs_roialign_bwd(_p(dpooled), _p(rois), _p(dfeat), rois.shape[0], B, Ntok, grid_h, grid_w, E, tok_off,
                                          self._stream()), "cs_roialign_bwd")

    def cosine_loss_fwd(self, student, teacher, stats, loss, weight):
        self._chk(student, teacher, stats, loss)
        K, E = student.shape
        self._ok(self.lib.cs_cosine_loss_fwd(_p(student), _p(teacher), _p(stats), _p(loss), K, E, weight, self._stream()),
                 "cs_cosine_loss_fwd")

    def cosine_loss_bwd(self, student, teacher, stats, dstudent, weight, grad_scale=1.0, upstream=None):
        self._chk(student, teacher, stats, dstudent, upstream)
        K, E = student.shape
        self._ok(self.lib.cs_cosine_loss_bwd(_p(student), _p(teacher), _p(stats), _p(dstudent), K, E, weight, grad_scale,
                                             _p(upstream), self._stream()), "cs_cosine_loss_bwd")

    def fed_bce_fwd(self, logits, tgt, rowloss, loss, ns, temp, weight):
        self._chk(logits, tgt, rowloss, loss)
        assert tgt.dtype == torch.int32
        self._ok(self.lib.cs_fed_bce_fwd(_p(logits), logits.stride(0), _p(tgt), _p(rowloss), _p(loss), logits.shape[0], ns, temp, weight,
                                         self._stream()), "cs_fed_bce_fwd")

    def fed_bce_bwd(self, logits, tgt, dz, ns, temp, weight, upstream=None):
        self._chk(logits, tgt, dz, upstream)
        self._ok(self.lib.cs_fed_bce_bwd(_p(logits), logits.stride(0), _p(tgt), _p(dz), dz.stride(0), logits.shape[0], ns, temp, weight,
                                         _p(upstream), self._stream()), "cs_fed_bce_bwd")

    def adamw_step(self, p, g, m, v, shadow, flags, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
        self._chk(p, g, m, v, shadow, flags)
        self._ok(self.lib.cs_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), _p(flags), p.numel(), lr, beta1, beta2, eps, wd,
                                        step, grad_scale, self._stream()), "cs_adamw_step")
