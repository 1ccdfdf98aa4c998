"""cuda-winograd_amd -- MI355X-native fused Winograd conv(+BN+ReLU) and 1x1-conv GEMM.

Python side of the drop-in boundary: a ctypes binding of the C-ABI declared in
``include/winograd_mi355x.h`` (the same shared library the C ``./Test`` driver
links), plus thin operator wrappers that take torch tensors.  PyTorch is only
plumbing here (device memory, streams, ``torch.distributed``); every compute
call goes to the hand-written HIP kernels in ``csrc/``.  There is NO CPU or
eager fallback: if the library is missing or no GPU is visible, calls raise.

The directory name contains a hyphen, so import it through
``__graft_entry__.load_package()`` (registers it as ``cuda_winograd_amd``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_long, c_size_t, c_void_p, POINTER

import torch  # imported first on purpose: the library then binds to torch's HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwinograd_mi355x.so")

# every symbol include/*.h declares (tests/test_abi.py checks the export table against it)
ABI_SYMBOLS = [
    "wino_abi_version", "wino_last_error_string", "wino_device_count", "wino_set_device",
    "wino_device_name", "wino_malloc", "wino_free", "wino_memset", "wino_memcpy_h2d",
    "wino_memcpy_d2h", "wino_memcpy_d2d", "wino_device_synchronize", "wino_stream_create",
    "wino_stream_destroy", "wino_stream_synchronize", "wino_event_create", "wino_event_destroy",
    "wino_event_record", "wino_event_elapsed_ms", "wino_filter_f2_elems", "wino_filter_f2_index",
    "wino_filter_transform_f2", "wino_filter_import_f4", "wino_conv3x3_bn_relu", "wino_conv3x3_prepare",
    "wino_conv3x3_bn_relu_hw", "wino_conv3x3_prepare_hw", "wino_conv3x3_direct_hw", "wino_conv3x3_plan",
    "wino_conv1x1_prepare", "wino_conv1x1_plan", "wino_conv1x1_bn_ex_hw", "wino_residual_block_hw",
    "wino_residual_block_workspace_bytes_hw",
    "wino_conv3x3_f4_bn_relu", "wino_conv3x3_f4_workspace_bytes",
    "wino_conv3x3_direct", "wino_conv1x1_bn", "wino_conv1x1_bn_ex", "wino_conv1x1_direct",
    "wino_residual_block", "wino_residual_block_workspace_bytes", "wino_driver_set_batch",
    "wino_driver_set_gpus", "wino_driver_set_quiet", "wino_driver_get_batch",
    "wino_driver_get_gpus", "wino_driver_last_result", "wino_driver_last_output", "wino_driver_pack_times",
    "wino_driver_set_gpu_alias", "wino_driver_set_stdout_compat", "wino_driver_get_stdout_compat",
    "wino_driver_cpu_baseline", "wino_last_status_name", "wino_debug_reload_knobs",
    "wino_residual_block_prepare", "wino_residual_block_prepare_hw", "wino_diag_conv3x3_clock",
    "wino_debug_tickets_in_use", "wino_stream_check", "wino_stream_reset_scratch", "wino_debug_poison_ticket",
    "wino_diag_last_clock", "wino_conv3x3_small_plan", "wino_conv1x1_small_plan", "wino_conv3x3_plan_groups", "wino_conv1x1_small_plan2",
    "wino_conv3x3_small_plan2", "wino_debug_conv1x1_models",
    # reference entry points + helpers (Kernel*.h, util.h)
    "kernel_128", "kernel_256", "kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in",
    "kernel_256_1_out", "get_parameter", "transpose", "getTimeMicroseconds64", "output_checker",
    "output_checker_accumulate",
]


class WinoError(RuntimeError):
    pass


class DriverResult(ctypes.Structure):
    _fields_ = [("mine_us", ctypes.c_double), ("comparator_us", ctypes.c_double),
                ("max_abs_err", ctypes.c_double), ("max_rel_err", ctypes.c_double),
                ("error_cnt", c_long), ("flops", ctypes.c_double), ("N", c_int), ("gpus", c_int),
                ("steady_us", ctypes.c_double)]


class CpuBaselineResult(ctypes.Structure):
    _fields_ = [("us", ctypes.c_double), ("gflops", ctypes.c_double), ("threads", c_int), ("reps", c_int),
                ("max_abs_diff", ctypes.c_double), ("max_rel_diff", ctypes.c_double)]


_lib = None


def lib() -> ctypes.CDLL:
    """Load libwinograd_mi355x.so (built in-tree by `make` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WinoError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "There is no fallback path.")
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    fp = c_void_p
    L.wino_last_error_string.restype = c_char_p
    L.wino_filter_f2_elems.restype = c_size_t
    L.wino_filter_f2_elems.argtypes = [c_int, c_int]
    L.wino_filter_f2_index.restype = c_long
    L.wino_filter_f2_index.argtypes = [c_int] * 5
    L.wino_filter_transform_f2.argtypes = [fp, fp, c_int, c_int, c_void_p]
    L.wino_filter_import_f4.argtypes = [fp, fp, c_int, c_int, c_void_p]
    L.wino_conv3x3_bn_relu.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_conv3x3_prepare.argtypes = [c_int, c_int, c_int, c_void_p]
    L.wino_conv1x1_prepare.argtypes = [c_long, c_int, c_int, c_void_p]
    L.wino_conv1x1_plan.argtypes = [c_long, c_int, c_int, c_int] + [POINTER(c_int)] * 5
    L.wino_conv3x3_bn_relu_hw.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_conv3x3_prepare_hw.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_conv3x3_f4_workspace_bytes.restype = c_size_t
    L.wino_conv3x3_f4_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.wino_conv3x3_f4_bn_relu.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, fp, c_size_t, c_void_p]
    L.wino_conv3x3_plan.argtypes = [c_int] * 6 + [POINTER(c_int), POINTER(c_int), POINTER(c_long), POINTER(c_int)]
    L.wino_conv3x3_direct_hw.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_conv3x3_direct.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_conv1x1_bn.argtypes = [fp, fp, fp, fp, fp, c_long, c_int, c_int, c_int, c_void_p]
    L.wino_conv1x1_direct.argtypes = [fp, fp, fp, fp, fp, c_long, c_int, c_int, c_int, c_void_p]
    L.wino_conv1x1_bn_ex.argtypes = [fp, fp, fp, fp, fp, fp, c_long, c_int, c_int, c_int, c_void_p]
    L.wino_residual_block_workspace_bytes.restype = c_size_t
    L.wino_residual_block_workspace_bytes.argtypes = [c_int, c_int]
    L.wino_residual_block.argtypes = [fp] * 11 + [c_int, c_int, c_int, fp, c_size_t, c_void_p]
    L.wino_conv1x1_bn_ex_hw.argtypes = [fp, fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.wino_residual_block_workspace_bytes_hw.restype = c_size_t
    L.wino_residual_block_workspace_bytes_hw.argtypes = [c_int, c_int, c_int, c_int]
    L.wino_residual_block_hw.argtypes = [fp] * 11 + [c_int] * 5 + [fp, c_size_t, c_void_p]
    L.wino_device_count.argtypes = [POINTER(c_int)]
    L.wino_driver_last_result.argtypes = [POINTER(DriverResult)]
    L.wino_driver_set_batch.argtypes = [c_int]
    L.wino_driver_set_gpus.argtypes = [c_int]
    L.wino_driver_set_quiet.argtypes = [c_int]
    L.wino_driver_set_gpu_alias.argtypes = [c_int]
    L.wino_driver_set_stdout_compat.argtypes = [c_int]
    L.wino_driver_pack_times.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    L.wino_driver_last_output.restype = POINTER(ctypes.c_float)
    L.wino_driver_last_output.argtypes = [POINTER(c_size_t)]
    L.wino_driver_cpu_baseline.argtypes = [POINTER(CpuBaselineResult)]
    L.wino_last_status_name.restype = c_char_p
    L.wino_residual_block_prepare.argtypes = [c_int, c_int, c_int, c_void_p]
    L.wino_residual_block_prepare_hw.argtypes = [c_int] * 5 + [c_void_p]
    L.wino_diag_conv3x3_clock.argtypes = [fp] * 5 + [c_int] * 3 + [fp, POINTER(c_int), c_void_p]
    L.wino_stream_check.argtypes = [c_void_p]
    L.wino_stream_reset_scratch.argtypes = [c_void_p]
    L.wino_debug_poison_ticket.argtypes = [c_void_p, c_long, ctypes.c_uint]
    L.wino_diag_last_clock.argtypes = [c_int, c_void_p, POINTER(ctypes.c_ulonglong)]
    L.wino_conv3x3_small_plan.argtypes = [c_int] * 6 + [POINTER(c_int)] * 4
    L.wino_conv3x3_small_plan2.argtypes = [c_int] * 6 + [POINTER(c_int)] * 5
    L.wino_debug_conv1x1_models.argtypes = [c_long, c_int, c_int, c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double)]
    L.wino_conv1x1_small_plan.argtypes = [c_long, c_int, c_int, c_int] + [POINTER(c_int)] * 3
    L.wino_conv1x1_small_plan2.argtypes = [c_long, c_int, c_int, c_int] + [POINTER(c_int)] * 5
    L.wino_conv3x3_plan_groups.argtypes = [c_int] * 6 + [POINTER(c_int)] * 4
    for name in ("kernel_128", "kernel_256", "kernel_128_1_in", "kernel_128_1_out",
                 "kernel_256_1_in", "kernel_256_1_out"):
        getattr(L, name).restype = c_int
        getattr(L, name).argtypes = []
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise WinoError(f"{what} failed (rc={rc}): {lib().wino_last_error_string().decode()}")


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise WinoError(f"{name} must be a CUDA(HIP) tensor -- this framework has no CPU path")
    if t.dtype != torch.float32:
        raise WinoError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def tickets_in_use() -> int:
    """Diagnostic (tests): non-zero stream-K ticket counters of the current stream's scratch after a
    synchronise.  Every launch must leave 0."""
    n = c_long(0)
    _check(lib().wino_debug_tickets_in_use(_stream(), ctypes.byref(n)), "wino_debug_tickets_in_use")
    return int(n.value)


def stream_check() -> None:
    """Waits for the current stream; raises WinoError (WINO_E_STATE) when its library-owned scratch cannot
    be trusted (an earlier launch failed, or a kernel found a ticket counter that was not zero at launch)."""
    _check(lib().wino_stream_check(_stream()), "wino_stream_check")


def stream_reset_scratch() -> None:
    """Recovery: zero the current stream's ticket counters and clear its error state."""
    _check(lib().wino_stream_reset_scratch(_stream()), "wino_stream_reset_scratch")


def poison_ticket(index: int, value: int) -> None:
    """Test hook: overwrite one ticket counter of the current stream's scratch."""
    _check(lib().wino_debug_poison_ticket(_stream(), int(index), int(value)), "wino_debug_poison_ticket")


def last_clock_ghz(kernel: int = 0):
    """The clock the chip held inside the MOST RECENT launch of a product kernel (0 = fused 3x3, 1 = 1x1 GEMM)
    on the current device: (GHz, shader cycles, microseconds) between workgroup 0's entry and exit stamps,
    or None when no launch has stamped yet.  Synchronises the current stream."""
    st = (ctypes.c_ulonglong * 4)()
    _check(lib().wino_diag_last_clock(int(kernel), _stream(), st), "wino_diag_last_clock")
    cyc, ticks = st[2] - st[0], st[3] - st[1]
    if st[1] == 0 or ticks <= 0 or cyc <= 0:
        return None
    return cyc / ticks * 0.1, int(cyc), ticks / 100.0


def small_plan_3x3(N: int, C: int, K: int, cus: int = 256, H: int = 14, W: int = 14):
    """(use, point_rows, split, workgroups) of the 3x3 latency kernel for this shape (host-side)."""
    v = [c_int(0) for _ in range(4)]
    _check(lib().wino_conv3x3_small_plan(N, H, W, C, K, cus, *[ctypes.byref(x) for x in v]), "wino_conv3x3_small_plan")
    return tuple(int(x.value) for x in v)


def small_plan_3x3_full(N: int, C: int, K: int, cus: int = 256, H: int = 14, W: int = 14):
    """(use, point_rows, split, col_tiles, workgroups) of the 3x3 latency kernel (host-side)."""
    v = [c_int(0) for _ in range(5)]
    _check(lib().wino_conv3x3_small_plan2(N, H, W, C, K, cus, *[ctypes.byref(x) for x in v]), "wino_conv3x3_small_plan2")
    return tuple(int(x.value) for x in v)


def small_plan_1x1(M: int, Cin: int, Kout: int, cus: int = 256):
    """(use, k_split, workgroups) of the 1x1 latency form for a plain layer of this shape (host-side)."""
    v = [c_int(0) for _ in range(3)]
    _check(lib().wino_conv1x1_small_plan(M, Cin, Kout, cus, *[ctypes.byref(x) for x in v]), "wino_conv1x1_small_plan")
    return tuple(int(x.value) for x in v)


def small_plan_1x1_full(M: int, Cin: int, Kout: int, cus: int = 256):
    """(use, k_split, row_tiles, col_tiles, workgroups) of the 1x1 latency form (host-side)."""
    v = [c_int(0) for _ in range(5)]
    _check(lib().wino_conv1x1_small_plan2(M, Cin, Kout, cus, *[ctypes.byref(x) for x in v]), "wino_conv1x1_small_plan2")
    return tuple(int(x.value) for x in v)


def _on_current_device(*tensors) -> None:
    """The library launches on the CURRENT device's stream: every tensor must live there."""
    cur = torch.cuda.current_device()
    for t in tensors:
        if t is not None and t.device.index != cur:
            raise WinoError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                            "wrap the call in torch.cuda.device(...)")


def _out(t: torch.Tensor, shape, name: str) -> torch.Tensor:
    """A caller-supplied output / workspace: written by the kernel as is, so it must be exactly what
    the kernel assumes (a wrong-sized or strided buffer would be an out-of-bounds GPU write)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise WinoError(f"{name} must be a float32 CUDA(HIP) tensor")
    if not t.is_contiguous():
        raise WinoError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise WinoError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


# --------------------------------------------------------------------------- operators
def filter_transform_f2(w_kcrs: torch.Tensor) -> torch.Tensor:
    """[K][C][3][3] taps -> packed F(2x2,3x3) filter buffer (opaque layout, 16*C*K floats)."""
    w = _dev(w_kcrs, "w_kcrs")
    K, C = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3):
        raise WinoError("w_kcrs must be [K][C][3][3]")
    U = torch.empty(lib().wino_filter_f2_elems(C, K), dtype=torch.float32, device=w.device)
    _check(lib().wino_filter_transform_f2(w.data_ptr(), U.data_ptr(), C, K, _stream()),
           "wino_filter_transform_f2")
    return U


def filter_import_f4(u36: torch.Tensor) -> torch.Tensor:
    """The reference's weight_winograd_C_K.bin tensor [36][C][K] -> packed F(2x2) buffer."""
    u = _dev(u36, "u36")
    if u.dim() != 3 or u.shape[0] != 36:
        raise WinoError("u36 must be [36][C][K]")
    C, K = int(u.shape[1]), int(u.shape[2])
    U = torch.empty(lib().wino_filter_f2_elems(C, K), dtype=torch.float32, device=u.device)
    _check(lib().wino_filter_import_f4(u.data_ptr(), U.data_ptr(), C, K, _stream()),
           "wino_filter_import_f4")
    return U


def conv3x3_bn_relu(inp: torch.Tensor, U: torch.Tensor, bn_bias: torch.Tensor,
                    bn_scale: torch.Tensor, relu: bool = True,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    """inp [N][H+2][W+2][C] -> out [N][H+2][W+2][K] (interior H x W, zero ring).  One HIP launch.
    [N][16][16][C] is the reference's 14x14 stage (wino_conv3x3_bn_relu); any other even H, W goes
    through wino_conv3x3_bn_relu_hw."""
    x = _dev(inp, "inp")
    U = _dev(U, "U")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    if x.dim() != 4 or x.shape[1] < 3 or x.shape[2] < 3:
        raise WinoError("inp must be [N][H+2][W+2][C]")
    N, Hp, Wp, C, K = int(x.shape[0]), int(x.shape[1]), int(x.shape[2]), int(x.shape[3]), int(b.numel())
    if U.numel() != 16 * C * K or s.numel() != K:
        raise WinoError("U / bn vectors do not match C, K")
    if out is None:
        out = torch.empty((N, Hp, Wp, K), dtype=torch.float32, device=x.device)
    else:
        _out(out, (N, Hp, Wp, K), "out")
    _on_current_device(x, U, b, s, out)
    if Hp == 16 and Wp == 16:
        _check(lib().wino_conv3x3_bn_relu(x.data_ptr(), U.data_ptr(), b.data_ptr(), s.data_ptr(),
                                          out.data_ptr(), N, C, K, int(relu), _stream()),
               "wino_conv3x3_bn_relu")
    else:
        _check(lib().wino_conv3x3_bn_relu_hw(x.data_ptr(), U.data_ptr(), b.data_ptr(), s.data_ptr(),
                                             out.data_ptr(), N, Hp - 2, Wp - 2, C, K, int(relu), _stream()),
               "wino_conv3x3_bn_relu_hw")
    return out


def conv3x3_f4_bn_relu(inp: torch.Tensor, u36: torch.Tensor, bn_bias: torch.Tensor, bn_scale: torch.Tensor,
                       relu: bool = True) -> torch.Tensor:
    """F(4x4,3x3) compatibility path: the reference's three stages on its own weight_winograd tensor
    u36 [36][C][K], consumed as is.  inp [N][16][16][C] -> out [N][16][16][K]."""
    x, u = _dev(inp, "inp"), _dev(u36, "u36")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    if x.dim() != 4 or tuple(x.shape[1:3]) != (16, 16) or u.dim() != 3 or u.shape[0] != 36:
        raise WinoError("inp must be [N][16][16][C], u36 [36][C][K]")
    N, C, K = int(x.shape[0]), int(x.shape[3]), int(u.shape[2])
    if int(u.shape[1]) != C or b.numel() != K or s.numel() != K:
        raise WinoError("u36 / bn vectors do not match C, K")
    out = torch.empty((N, 16, 16, K), dtype=torch.float32, device=x.device)
    nbytes = lib().wino_conv3x3_f4_workspace_bytes(N, C, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(lib().wino_conv3x3_f4_bn_relu(x.data_ptr(), u.data_ptr(), b.data_ptr(), s.data_ptr(), out.data_ptr(),
                                         N, C, K, int(relu), ws.data_ptr(), nbytes, _stream()),
           "wino_conv3x3_f4_bn_relu")
    return out


def conv3x3_prepare(N: int, C: int, K: int, H: int = 14, W: int = 14) -> None:
    """Allocate the library-owned stream-K scratch of conv3x3_bn_relu for this shape on the current
    device and stream ahead of time (needed before capturing the call into a HIP graph)."""
    _check(lib().wino_conv3x3_prepare_hw(int(N), int(H), int(W), int(C), int(K), _stream()),
           "wino_conv3x3_prepare_hw")


def conv3x3_direct(inp, w_kcrs, bn_bias, bn_scale, relu: bool = True) -> torch.Tensor:
    """Comparator: direct 3x3 conv + BN + ReLU on the GPU (not the product path); any H, W."""
    x, w = _dev(inp, "inp"), _dev(w_kcrs, "w_kcrs")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    N, Hp, Wp, C, K = int(x.shape[0]), int(x.shape[1]), int(x.shape[2]), int(x.shape[3]), int(w.shape[0])
    out = torch.empty((N, Hp, Wp, K), dtype=torch.float32, device=x.device)
    _check(lib().wino_conv3x3_direct_hw(x.data_ptr(), w.data_ptr(), b.data_ptr(), s.data_ptr(),
                                        out.data_ptr(), N, Hp - 2, Wp - 2, C, K, int(relu), _stream()),
           "wino_conv3x3_direct_hw")
    return out


def conv1x1_bn(A: torch.Tensor, B: torch.Tensor, bn_bias: torch.Tensor, bn_scale: torch.Tensor,
               relu: bool, out: torch.Tensor | None = None) -> torch.Tensor:
    """A [M][Cin] @ B [Cin][Kout] -> scale*(.)+bias (+ReLU), [M][Kout].  One HIP launch."""
    a, bm = _dev(A, "A"), _dev(B, "B")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    if a.dim() != 2 or bm.dim() != 2 or a.shape[1] != bm.shape[0]:
        raise WinoError("A must be [M][Cin], B [Cin][Kout]")
    M, Cin, Kout = int(a.shape[0]), int(a.shape[1]), int(bm.shape[1])
    if b.numel() != Kout or s.numel() != Kout:
        raise WinoError("bn vectors do not match Kout")
    if out is None:
        out = torch.empty((M, Kout), dtype=torch.float32, device=a.device)
    else:
        _out(out, (M, Kout), "out")
    _on_current_device(a, bm, b, s, out)
    _check(lib().wino_conv1x1_bn(a.data_ptr(), bm.data_ptr(), b.data_ptr(), s.data_ptr(),
                                 out.data_ptr(), M, Cin, Kout, int(relu), _stream()),
           "wino_conv1x1_bn")
    return out


RELU, A_PADDED, C_PADDED, ADD_RESIDUAL = 1, 2, 4, 8  # WINO_* flag bits of wino_conv1x1_bn_ex


def residual_block_prepare(N: int, C4: int, Cm: int, H: int = 14, W: int = 14) -> None:
    """Allocate the scratch of residual_block's three launches for the current stream (before graph capture)."""
    _check(lib().wino_residual_block_prepare_hw(int(N), int(H), int(W), int(C4), int(Cm), _stream()),
           "wino_residual_block_prepare_hw")


def conv3x3_clock_ghz(inp, U, bn_bias, bn_scale, out) -> float:
    """Diagnostic: one launch of the 3x3 throughput kernel's stamped build (14x14 only); returns the
    median over workgroups of the in-kernel clock, d(s_memtime) / d(s_memrealtime) * 0.1 GHz."""
    x, U = _dev(inp, "inp"), _dev(U, "U")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    N, C, K = int(x.shape[0]), int(x.shape[3]), int(b.numel())
    _out(out, (N, 16, 16, K), "out")
    _on_current_device(x, U, b, s, out)
    stamps = torch.zeros(4 * 2048, dtype=torch.int64, device=x.device)
    wgs = c_int(0)
    _check(lib().wino_diag_conv3x3_clock(x.data_ptr(), U.data_ptr(), b.data_ptr(), s.data_ptr(), out.data_ptr(),
                                         N, C, K, stamps.data_ptr(), ctypes.byref(wgs), _stream()),
           "wino_diag_conv3x3_clock")
    st = stamps[:4 * wgs.value].view(-1, 4).cpu()      # {cycles, 100 MHz ticks} at start and at end
    cyc, ticks = (st[:, 2] - st[:, 0]).double(), (st[:, 3] - st[:, 1]).double()
    ok = (ticks > 0) & (st[:, 1] != 0)
    if not bool(ok.any()):
        raise WinoError("clock probe returned no stamps")
    return float((cyc[ok] / ticks[ok]).median()) * 0.1


def conv1x1_prepare(M: int, Cin: int, Kout: int) -> None:
    """Allocate the 1x1 layer's stream-K scratch for the current stream (before graph capture)."""
    _check(lib().wino_conv1x1_prepare(int(M), int(Cin), int(Kout), _stream()), "wino_conv1x1_prepare")


def conv1x1_bn_ex(A, B, bn_bias, bn_scale, flags: int, residual=None, out=None, hw=None) -> torch.Tensor:
    """Chaining form of the 1x1 layer: A and/or C may be the padded [N][H+2][W+2][.] tensors of the
    3x3 layer (flags A_PADDED / C_PADDED), a residual [M][Kout] may be added before the ReLU.
    The feature-map size comes from the padded A, else from a 4-D unpadded A [N][H][W][Cin], else
    from `hw`, else it is the reference's 14 x 14 (wino_conv1x1_bn_ex); anything but 14 x 14 goes
    through wino_conv1x1_bn_ex_hw."""
    a, bm = _dev(A, "A"), _dev(B, "B")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    Cin, Kout = int(bm.shape[0]), int(bm.shape[1])
    H = W = None
    if flags & A_PADDED:
        if a.dim() != 4 or int(a.shape[3]) != Cin or a.shape[1] < 3 or a.shape[2] < 3:
            raise WinoError("A_PADDED: A must be [N][H+2][W+2][Cin]")
        H, W = int(a.shape[1]) - 2, int(a.shape[2]) - 2
        N = int(a.shape[0])
        M = N * H * W
    else:
        if a.dim() == 4:
            H, W = int(a.shape[1]), int(a.shape[2])
        a = a.reshape(-1, Cin)
        M = int(a.shape[0])
    if hw is not None:
        if H is not None and (H, W) != tuple(hw):
            raise WinoError(f"hw={tuple(hw)} contradicts A's {H}x{W}")
        H, W = int(hw[0]), int(hw[1])
    if H is None:
        H = W = 14
    padded = bool(flags & (A_PADDED | C_PADDED))
    if padded and M % (H * W):
        raise WinoError(f"padded layouts need M = N*{H}*{W}, got M={M}")
    r = _dev(residual, "residual") if residual is not None else None
    if r is not None and r.numel() != M * Kout:
        raise WinoError(f"residual must hold M*Kout = {M * Kout} values, got {r.numel()}")
    shape = (M // (H * W), H + 2, W + 2, Kout) if flags & C_PADDED else (M, Kout)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=a.device)
    elif out.numel() != M * Kout or flags & C_PADDED:
        _out(out, shape, "out")
    else:
        _out(out, None, "out")   # unpadded: any contiguous view of M*Kout values ([M][Kout] or [N][H][W][Kout])
    _on_current_device(a, bm, b, s, r, out)
    if padded and (H, W) != (14, 14):
        _check(lib().wino_conv1x1_bn_ex_hw(a.data_ptr(), bm.data_ptr(), b.data_ptr(), s.data_ptr(),
                                           r.data_ptr() if r is not None else None, out.data_ptr(),
                                           M // (H * W), H, W, Cin, Kout, int(flags), _stream()),
               "wino_conv1x1_bn_ex_hw")
    else:
        _check(lib().wino_conv1x1_bn_ex(a.data_ptr(), bm.data_ptr(), b.data_ptr(), s.data_ptr(),
                                        r.data_ptr() if r is not None else None, out.data_ptr(),
                                        M, Cin, Kout, int(flags), _stream()), "wino_conv1x1_bn_ex")
    return out


def residual_block(x, w1, bn1, U2, bn2, w3, bn3, out=None, workspace=None) -> torch.Tensor:
    """ResNet bottleneck: x [N][H][W][C4] -> same shape (the reference's stage is 14 x 14:
    wino_residual_block; other sizes: wino_residual_block_hw).  bnX = (bias, scale) folded BN
    vectors; w1 [C4][Cm], w3 [Cm][C4]; U2 from filter_transform_f2 (Cm -> Cm)."""
    x = _dev(x, "x")
    if x.dim() != 4:
        raise WinoError("x must be [N][H][W][C4]")
    N, H, W, C4 = (int(v) for v in x.shape)
    w1, w3, U2 = _dev(w1, "w1"), _dev(w3, "w3"), _dev(U2, "U2")
    Cm = int(w1.shape[1])
    vecs = [_dev(v, "bn") for pair in (bn1, bn2, bn3) for v in pair]
    need = lib().wino_residual_block_workspace_bytes_hw(N, H, W, Cm)
    if workspace is None:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    else:
        _out(workspace, None, "workspace")
        if workspace.numel() * 4 < need:
            raise WinoError(f"workspace too small: {workspace.numel() * 4} bytes, need {need}")
    if out is None:
        out = torch.empty_like(x)
    else:
        _out(out, x.shape, "out")
    _on_current_device(x, w1, w3, U2, out, workspace, *vecs)
    args = (x.data_ptr(), w1.data_ptr(), vecs[0].data_ptr(), vecs[1].data_ptr(),
            U2.data_ptr(), vecs[2].data_ptr(), vecs[3].data_ptr(),
            w3.data_ptr(), vecs[4].data_ptr(), vecs[5].data_ptr(), out.data_ptr())
    if (H, W) == (14, 14):
        _check(lib().wino_residual_block(*args, N, C4, Cm, workspace.data_ptr(), workspace.numel() * 4, _stream()),
               "wino_residual_block")
    else:
        _check(lib().wino_residual_block_hw(*args, N, H, W, C4, Cm, workspace.data_ptr(),
                                            workspace.numel() * 4, _stream()), "wino_residual_block_hw")
    return out


def conv1x1_direct(A, B, bn_bias, bn_scale, relu: bool) -> torch.Tensor:
    a, bm = _dev(A, "A"), _dev(B, "B")
    b, s = _dev(bn_bias, "bn_bias"), _dev(bn_scale, "bn_scale")
    M, Cin, Kout = int(a.shape[0]), int(a.shape[1]), int(bm.shape[1])
    out = torch.empty((M, Kout), dtype=torch.float32, device=a.device)
    _check(lib().wino_conv1x1_direct(a.data_ptr(), bm.data_ptr(), b.data_ptr(), s.data_ptr(),
                                     out.data_ptr(), M, Cin, Kout, int(relu), _stream()),
           "wino_conv1x1_direct")
    return out


# ---------------------------------------------------------------- batch split (multi-GPU)
def shard_range(N: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous image range [n0, n1) of `rank` when N images are split over `world` GPUs.
    The path has no exchange step: every image is independent, weights are replicated, so
    there is no collective on the data path."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (N * rank) // world, (N * (rank + 1)) // world
