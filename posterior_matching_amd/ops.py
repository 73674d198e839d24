"""Thin typed wrappers from torch device tensors to the C ABI (include/pmhip.h).

PyTorch is used for device memory and streams only; every arithmetic op of the step runs in
libpmhip.so.  All calls enqueue on torch's current stream and never synchronise.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_RELU, AUX_AFTER_RES, LEAKY_SLOPE, GatherDesc  # noqa: F401


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expects contiguous f32 device tensors"
    return t.data_ptr()


class KernelTimer:
    """Brackets every C-ABI launch with HIP events on the launching stream (bench.py's live
    per-kernel timing; never active in a captured graph)."""

    def __init__(self):
        self.records = []   # (tag, work, start_event, stop_event)

    def per_call(self):
        """[(tag, detail, ms, flops)] in launch order"""
        rows = []
        for tag, work, e0, e1 in self.records:
            e1.synchronize()
            rows.append((tag, work.get("detail", ""), e0.elapsed_ms(e1), work.get("flops", 0.0)))
        return rows

    def summary(self, robust: bool = True):
        """per kernel name: calls, total ms, algorithmic flops / bytes.  robust: a launch whose event-to-event time exceeds
        3x the median of its (name, shape) group is counted at that median - an event pair also spans whatever delayed
        the launch (one 400 us reading among 26 us ones was seen on an otherwise idle stream); `outliers` says how many"""
        rows = []
        for tag, work, e0, e1 in self.records:
            e1.synchronize()
            rows.append((tag, work, e0.elapsed_ms(e1)))
        med = {}
        if robust:
            groups = {}
            for tag, work, ms in rows:
                groups.setdefault((tag, work.get("detail", "")), []).append(ms)
            med = {k: sorted(v)[len(v) // 2] for k, v in groups.items()}
        out = {}
        for tag, work, ms in rows:
            r = out.setdefault(tag, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "outliers": 0})
            m = med.get((tag, work.get("detail", "")))
            if m is not None and ms > 3.0 * m:
                ms = m
                r["outliers"] += 1
            r["calls"] += 1
            r["ms"] += ms
            r["flops"] += work.get("flops", 0.0)
            r["bytes"] += work.get("bytes", 0.0)
        return out


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t
    _lib.load().pm_kernel_names_enable(1 if t is not None else 0)


# ---- launch plans: the host side of a training step recorded once, replayed without the Python model code ------
_recording: Optional[list] = None


class LaunchPlan:
    """The C-ABI calls (bound function + final argument tuple, stream handle included) and the cross-stream
    synchronisation calls of one step, in issue order.  Every buffer of a step is static, so replaying the list
    reproduces the step; it keeps the multi-stream overlap that a captured HIP graph loses on ROCm 7.2 and costs
    ~2 us of host time per launch instead of ~11 us through the model classes."""

    def __init__(self, calls):
        self.calls = calls

    def __len__(self):
        return len(self.calls)

    def replay(self) -> None:
        for fn, args, name in self.calls:
            rc = fn(*args)
            if type(rc) is int and rc:          # C-ABI status; stream / collective calls return None or a tensor
                _lib.check(rc, name)


def begin_recording() -> None:
    global _recording
    if _timer is not None:
        raise RuntimeError("cannot record a launch plan while a KernelTimer is active")
    _recording = []


def end_recording() -> LaunchPlan:
    global _recording
    plan, _recording = LaunchPlan(_recording or []), None
    return plan


def wait_stream(waiter, other) -> None:
    """waiter.wait_stream(other), recorded into the active launch plan"""
    if waiter is other or waiter == other:
        return
    waiter.wait_stream(other)
    if _recording is not None:
        _recording.append((waiter.wait_stream, (other,), "wait_stream"))


def record_event(event, stream) -> None:
    event.record(stream)
    if _recording is not None:
        _recording.append((event.record, (stream,), "event.record"))


def wait_event(stream, event) -> None:
    stream.wait_event(event)
    if _recording is not None:
        _recording.append((stream.wait_event, (event,), "wait_event"))


def host_call(fn, *args) -> None:
    """a host-side call that belongs to the step (the gradient all-reduce), recorded into the active plan"""
    fn(*args)
    if _recording is not None:
        _recording.append((fn, args, getattr(fn, "__name__", "host_call")))


# ---- kernel coverage: which kernel variants a piece of host code launched (tests/test_gpu_zz_coverage.py) -------------
_coverage: Optional[set] = None


def coverage_begin() -> None:
    """from here on every C-ABI call adds "<kernel name as rocprofv3 prints it>[<variant>]" (entry points with one kernel:
    the entry point's name) to a set; eager calls only (a replayed launch plan repeats what was recorded)"""
    global _coverage
    _coverage = set()
    _lib.load().pm_kernel_names_enable(1)


def coverage_take() -> set:
    """the kernel variants recorded since coverage_begin() / the previous take; recording goes on with an empty set"""
    global _coverage
    out, _coverage = (_coverage or set()), (set() if _coverage is not None else None)
    return out


def coverage_end() -> set:
    global _coverage
    out, _coverage = _coverage or set(), None
    if _timer is None:
        _lib.load().pm_kernel_names_enable(0)
    return out


# ---- device-side time stamps behind every launch (tools/stamp_timeline.py) -----------------------------------------------
_stamps: Optional[dict] = None


def stamps_begin(capacity: int = 4096) -> None:
    """from here on every C-ABI launch is followed, on its stream, by a one-thread kernel that stores the device's real-time
    counter (pm_stamp) into the next slot of a buffer; a launch plan recorded meanwhile replays the stamps too"""
    global _stamps
    _stamps = {"buf": torch.zeros(capacity, dtype=torch.int64, device="cuda"), "names": [], "streams": []}
    _lib.load().pm_kernel_names_enable(1)


def stamps_end():
    """[(kernel name, stream handle, counter value in 10 ns ticks)] in issue order"""
    global _stamps
    st, _stamps = _stamps, None
    if _timer is None and _coverage is None:
        _lib.load().pm_kernel_names_enable(0)
    torch.cuda.synchronize()
    vals = st["buf"].cpu().tolist()
    return [(n, s, vals[i]) for i, (n, s) in enumerate(zip(st["names"], st["streams"]))]


def _stamp(lib, stream: int, name: str) -> None:
    i = len(_stamps["names"])
    if i >= _stamps["buf"].numel():
        return
    _stamps["names"].append(name)
    _stamps["streams"].append(stream)
    args = (stream, _stamps["buf"].data_ptr() + 8 * i)
    _lib.check(lib.pm_stamp(*args), "pm_stamp")
    if _recording is not None:
        _recording.append((lib.pm_stamp, args, "pm_stamp"))


def _call(fname: str, *args, tag: Optional[str] = None, work: Optional[dict] = None) -> None:
    lib = _lib.load()
    fn = getattr(lib, fname)
    if _timer is None:
        full = (_stream(),) + args
        if _coverage is not None or _stamps is not None:
            lib.pm_clear_kernel_name()
            _lib.check(fn(*full), fname)
            var = lib.pm_last_kernel_variant().decode()
            name = (lib.pm_last_kernel_name().decode() or fname) + (f"[{var}]" if var else "")
            if _coverage is not None:
                _coverage.add(name)
        else:
            _lib.check(fn(*full), fname)
        if _recording is not None:
            _recording.append((fn, full, fname))
        if _stamps is not None:
            _stamp(lib, full[0], name)
        return
    e0, e1 = Event(), Event()
    lib.pm_clear_kernel_name()
    e0.record()
    _lib.check(fn(_stream(), *args), fname)
    e1.record()
    # the library reports which kernel variant it launched (as rocprofv3 names it): rows of the live table = rows of
    # the rocprof summary.  `tag` (the host-side mirror of the dispatch rules) only names what the library does not.
    name = lib.pm_last_kernel_name().decode()
    _timer.records.append((name or tag or fname, work or {}, e0, e1))


def _nbytes(*ts) -> float:
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


def _iptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int32
    return t.data_ptr()


# ------------------------------------------------------------------------------------------
# conv geometry: how each layer maps onto pm_gather_desc
# ------------------------------------------------------------------------------------------
def same_padding(in_size: int, k: int, s: int):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv_transpose_padding(k: int, s: int, padding: str):
    if padding == "SAME":
        pad_len = k + s - 2
        pad_a = k - 1 if s > k - 1 else int(math.ceil(pad_len / 2))
    elif padding == "VALID":
        pad_len = k + s - 2 + max(k - s, 0)
        pad_a = k - 1
    else:
        raise ValueError(padding)
    return pad_a, pad_len - pad_a


@dataclass
class LayerGeom:
    """A conv / transposed-conv / dense layer: x [B,IH,IW,CI] -> y [B,OH,OW,CO]."""

    kind: str  # "conv" | "convT" | "dense"
    IH: int
    IW: int
    CI: int
    OH: int
    OW: int
    CO: int
    k: int = 1
    s: int = 1
    pad: int = 0  # pad_lo (conv) or pad_a (convT)
    # masked convolution (PixelCNN `w *= mask`, reference pixel_cnn.py:392-422): the stored weight is
    # [full_kh, full_kw, CI, CO]; only rows [0, kh) x columns [0, kw) of it are walked
    kh: int = 0      # 0: square kernel k x k
    kw: int = 0
    pad_x: int = -1  # -1: same as pad
    full_kh: int = 0
    full_kw: int = 0

    @staticmethod
    def conv(ih, iw, ci, co, k, s, padding):
        if padding == "SAME":
            oh, ow = -(-ih // s), -(-iw // s)
            pad = same_padding(ih, k, s)[0]
            assert same_padding(iw, k, s)[0] == pad
        else:
            oh, ow, pad = (ih - k) // s + 1, (iw - k) // s + 1, 0
        return LayerGeom("conv", ih, iw, ci, oh, ow, co, k, s, pad)

    @staticmethod
    def conv_t(ih, iw, ci, co, k, s, padding):
        pa, pb = conv_transpose_padding(k, s, padding)
        oh = (ih - 1) * s + 1 + pa + pb - k + 1
        ow = (iw - 1) * s + 1 + pa + pb - k + 1
        return LayerGeom("convT", ih, iw, ci, oh, ow, co, k, s, pa)

    @staticmethod
    def dense(ci, co):
        return LayerGeom("dense", 1, 1, ci, 1, 1, co)

    @staticmethod
    def masked_conv(ih, iw, ci, co, full_kh, full_kw, valid_rows, valid_cols):
        """Stride-1 SAME conv whose kernel mask keeps rows [0, valid_rows) x columns [0, valid_cols)
        (every mask _make_kernel_constraint builds for num_hierarchies = 1 has this top-left form)."""
        py, px = same_padding(ih, full_kh, 1)[0], same_padding(iw, full_kw, 1)[0]
        return LayerGeom("conv", ih, iw, ci, ih, iw, co, k=0, s=1, pad=py, kh=valid_rows, kw=valid_cols, pad_x=px,
                         full_kh=full_kh, full_kw=full_kw)

    @property
    def KH(self):
        return self.kh or self.k

    @property
    def KW(self):
        return self.kw or self.k

    @property
    def weight_shape(self):
        if self.kind == "conv":
            return (self.full_kh or self.k, self.full_kw or self.k, self.CI, self.CO)
        if self.kind == "convT":
            return (self.k, self.k, self.CO, self.CI)
        return (self.CI, self.CO)

    def _desc(self, B: int, mode: str, groups=1, in_gs=0, w_gs=0, out_gs=0, bias_gs=0, w_ld=None) -> GatherDesc:
        """mode: 'fwd' (x -> y), 'dgrad' (dy -> dx) or 'wgrad' (same index rule as fwd).
        w_ld overrides the dense weight's row stride (grouped column slices of a wider matrix)."""
        g = self
        d = GatherDesc()
        d.B, d.KH, d.KW, d.groups = B, g.KH, g.KW, groups
        d.kws = g.full_kw or g.KW
        d.in_gs, d.w_gs, d.out_gs, d.bias_gs = in_gs, w_gs, out_gs, bias_gs
        if mode == "dgrad":  # the gathered operand is dy, the produced one dx
            d.in_gs, d.out_gs = out_gs, in_gs
        d.in_act = d.out_act = d.aux_act = ACT_NONE
        d.slope = LEAKY_SLOPE
        fwd_like = mode in ("fwd", "wgrad")
        if fwd_like:
            d.IH, d.IW, d.C, d.OH, d.OW, d.N = g.IH, g.IW, g.CI, g.OH, g.OW, g.CO
        else:
            d.IH, d.IW, d.C, d.OH, d.OW, d.N = g.OH, g.OW, g.CO, g.IH, g.IW, g.CI
        if g.kind in ("conv", "dense"):
            ld = w_ld if w_ld is not None else g.CO
            if fwd_like:
                d.a, d.cs, d.off, d.d = g.s, 1, -g.pad, 1
                d.wts, d.wcs, d.wns = g.CI * g.CO, ld, 1
            else:
                d.a, d.cs, d.off, d.d = 1, -1, g.pad, g.s
                d.wts, d.wcs, d.wns = g.CI * g.CO, 1, ld
        else:  # convT, weight [k,k,CO,CI]
            if fwd_like:
                d.a, d.cs, d.off, d.d = 1, 1, -g.pad, g.s
                d.wts, d.wcs, d.wns = g.CO * g.CI, 1, g.CI
            else:
                d.a, d.cs, d.off, d.d = g.s, -1, g.pad, 1
                d.wts, d.wcs, d.wns = g.CO * g.CI, g.CI, 1
        d.off_x = d.off
        if g.pad_x >= 0 and g.pad_x != g.pad:
            d.off_x = -g.pad_x if fwd_like else g.pad_x
        return d


def _valid_pairs(out_n: int, in_n: int, k: int, a: int, cs: int, off: int, d: int) -> int:
    """number of (output coordinate, tap) pairs whose source coordinate exists"""
    cnt = 0
    for p in range(out_n):
        for t in range(k):
            v = p * a + t * cs + off
            if v >= 0 and v % d == 0 and v // d < in_n:
                cnt += 1
    return cnt


def _algorithmic_flops(d: GatherDesc) -> float:
    """2 * MACs the layer needs: only (position, tap) pairs that touch a real input pixel count,
    so neither the zero-dilation taps of the stride-2 transposed / data-gradient forms nor the
    padding taps at the image border are claimed as work."""
    vy = _valid_pairs(d.OH, d.IH, d.KH, d.a, d.cs, d.off, d.d)
    vx = _valid_pairs(d.OW, d.IW, d.KW, d.a, d.cs, d.off_x, d.d)
    return 2.0 * d.B * vy * vx * d.C * d.N * d.groups


_MODES = {0: "TU1", 1: "TU2", 2: "V4", 3: "V1"}   # loader modes of csrc/pm_conv.hip


def _detail(d: GatherDesc) -> str:
    return (f"B{d.B} in{d.IH}x{d.IW}x{d.C} out{d.OH}x{d.OW}x{d.N} k{d.KH} a{d.a} d{d.d} g{d.groups}")


_SK_SCRATCH: dict = {}      # stream -> [float32 tensors]: slab scratch of the split-K GEMMs launched on that stream
SPLITK_SLABS = not os.environ.get("PM_SPLITK_ATOMIC")      # A/B knob: the f32-atomic split-K of rounds 1-3


def _splitk_scratch(desc: GatherDesc, bf16: bool, inp):
    """The slab scratch of a GEMM that splits K (pm_gemm_splitk_floats), None when it does not.  One buffer per stream, reused by
    every launch on it (stream order keeps a launch's epilogue ahead of the next launch's slices); buffers only ever grow by
    adding a new one, so a recorded LaunchPlan's pointers stay valid."""
    if not SPLITK_SLABS:
        return None
    n = C.c_longlong(0)
    if _lib.load().pm_gemm_splitk_floats(C.byref(desc), int(bf16), int(inp.data_ptr() % 16 == 0), C.byref(n)) or n.value <= 0:
        return None
    bufs = _SK_SCRATCH.setdefault((inp.device.index, _stream()), [])
    for t in bufs:
        if t.numel() >= n.value:
            return t
    size = 1 << max(18, (n.value - 1).bit_length())
    bufs.append(torch.empty(size, dtype=torch.float32, device=inp.device))
    return bufs[-1]


def gather_gemm(desc: GatherDesc, inp, w, bias, aux, res, out) -> None:
    tag = work = None
    if _timer is not None:
        bm, bn, vec = C.c_int(), C.c_int(), C.c_int()
        _lib.load().pm_query_gemm_plan(C.byref(desc), int(inp.data_ptr() % 16 == 0), C.byref(bm), C.byref(bn),
                                       C.byref(vec))
        if vec.value >= 16:
            tag = f"direct_gemm_kernel<{bn.value // 32},{_MODES[vec.value - 16]}>"
        else:
            tag = f"gather_gemm_kernel<{bm.value},{bn.value},{_MODES[vec.value]}>"
        work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(inp, w, aux, res, out), "detail": _detail(desc)}
    sk = _splitk_scratch(desc, False, inp)
    if sk is not None:
        _call("pm_gather_gemm_sk", C.byref(desc), _ptr(inp), _ptr(w), _ptr(bias), _ptr(aux), _ptr(res), _ptr(out),
              sk.data_ptr(), sk.numel(), tag=tag, work=work)
        return
    _call("pm_gather_gemm", C.byref(desc), _ptr(inp), _ptr(w), _ptr(bias), _ptr(aux), _ptr(res), _ptr(out), tag=tag,
          work=work)


def bf16_supported(desc: GatherDesc) -> bool:
    """pm_gather_gemm_bf16 preconditions (include/pmhip.h)."""
    return desc.C % 8 == 0 and desc.d in (1, 2) and desc.KH * desc.KW * ((desc.C + 31) // 32) <= 256


def _patch_form(d: GatherDesc) -> bool:
    """mirrors plan_patch (csrc/pm_conv.hip): stride-1 problems on grids >= 12 wide run the patch-staged kernel"""
    if d.groups != 1 or d.a != 1 or d.d != 1 or d.C % 32 != 0 or d.in_act != ACT_NONE:
        return False
    if d.KH * d.KW < 4 or d.OW < 12 or d.OH < 4:
        return False
    tw = 32 if d.OW > 16 else 16
    th = 128 // tw
    tiles = -(-d.OW // tw) * tw * -(-d.OH // th) * th
    if tiles * 2 > 3 * d.OH * d.OW:
        return False
    nb = 64 if d.N > 32 else 32
    lds = 2 * 2 * nb * 40 * 2 * 2 + (d.KH * d.KW * (d.C // 32) + 2) * 16 + (th + d.KH - 1) * (tw + d.KW - 1) * (d.C + 8) * 4
    return lds <= 150 * 1024


def _patch_cp_form(d: GatherDesc) -> bool:
    """mirrors plan_patch_cp (csrc/pm_conv.hip): deep stride-1 layers whose patch fits LDS a channel chunk at a time"""
    if d.groups != 1 or d.a != 1 or d.d != 1 or d.C % 64 != 0 or d.C <= 64 or d.in_act != ACT_NONE:
        return False
    if d.KH * d.KW < 4 or d.OW < 12 or d.OH < 4 or os.environ.get("PM_NO_PATCH_CP"):
        return False
    tw = 32 if d.OW > 16 else 16
    th = 128 // tw
    if -(-d.OW // tw) * tw * -(-d.OH // th) * th * 2 > 3 * d.OH * d.OW:
        return False
    return (th + d.KH - 1) * (tw + d.KW - 1) * 16 <= 12 * 256 and (d.KH * d.KW * (d.C // 32)) % 4 == 0


def _patch_d2_form(d: GatherDesc) -> bool:
    """mirrors plan_patch_d2 (csrc/pm_conv.hip): zero-dilated problems whose four residue classes share one staged patch"""
    if d.groups != 1 or d.a != 1 or d.d != 2 or d.C % 32 != 0 or d.in_act != ACT_NONE or d.cs not in (1, -1):
        return False
    if d.OH % 2 or d.OW % 2 or d.KH > 8 or d.KW > 8 or d.KH * d.KW < 4:
        return False
    ch, cw = d.OH // 2, d.OW // 2
    if ch > d.IH + 4 or cw > d.IW + 4:
        return False
    if cw > 8:
        tw, ni = 16, 1
    elif ch <= 8 and os.environ.get("PM_PATCH_D2_SMALL"):
        tw, ni = 8, 2
    else:
        return False
    th = 128 // (tw * ni)
    if -(-cw // tw) * tw * -(-ch // th) * th * 2 > 3 * ch * cw:
        return False
    nb = 64 if d.N > 32 else 32
    nsteps = d.KH * d.KW * (d.C // 32)
    lds = (2 * 2 * nb * 40 * 2 * 2 + 64 + ((nsteps + 3) & ~1) * 8
           + ni * (th + (d.KH + 1) // 2 + 1) * (tw + (d.KW + 1) // 2 + 1) * (d.C + 8) * 4)
    return nsteps <= 2048 and lds <= 150 * 1024


def gather_gemm_bf16(desc: GatherDesc, inp, wsplit, bias, aux, res, out, out2=None, act2=ACT_NONE, in_colsum=None) -> None:
    """out2 (optional): a second store out2 = act2(out) from the same launch (pm_gather_gemm_bf16_dual)
    in_colsum (optional, dgrad_insum_ok only): += the column sums of `inp` from the same launch (pm_gather_gemm_bf16_insum)"""
    tag = work = None
    if _timer is not None:
        dense = (desc.KH == desc.KW == 1 and desc.IH == desc.OH and desc.IW == desc.OW and desc.d == 1 and desc.a == 1
                 and desc.off == 0 and desc.off_x == 0)
        wgs64 = -(-desc.B * desc.OH * desc.OW // 128) * -(-desc.N // 64) * desc.groups
        rn = 2 if desc.N > 32 and wgs64 >= 512 else 1        # pm_gather_gemm_bf16: 32-column workgroups on short grids
        tag = f"direct_gemm_bf16_kernel<{rn}, {desc.d}, {desc.in_act}, {'true' if dense else 'false'}>"   # template args
        if _patch_form(desc):
            tag = f"patch_conv_bf16_kernel<{2 if desc.N > 32 else 1}>"
        elif _patch_cp_form(desc):
            tag = f"patch_conv_cp_bf16_kernel<{2 if desc.N > 32 else 1}>"
        elif _patch_d2_form(desc):
            tag = f"patch_d2_bf16_kernel<{2 if desc.N > 32 else 1}>"
        work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(inp, aux, res, out), "detail": _detail(desc)}
    sk = _splitk_scratch(desc, True, inp) if in_colsum is None else None
    if sk is not None:
        _call("pm_gather_gemm_bf16_sk", C.byref(desc), _ptr(inp), wsplit.data_ptr(), _ptr(bias), _ptr(aux), _ptr(res),
              _ptr(out), _ptr(out2), act2, sk.data_ptr(), sk.numel(), tag=tag, work=work)
        return
    if out2 is not None:
        assert in_colsum is None
        _call("pm_gather_gemm_bf16_dual", C.byref(desc), _ptr(inp), wsplit.data_ptr(), _ptr(bias), _ptr(aux), _ptr(res),
              _ptr(out), _ptr(out2), act2, tag=tag, work=work)
        return
    if in_colsum is not None:
        _call("pm_gather_gemm_bf16_insum", C.byref(desc), _ptr(inp), wsplit.data_ptr(), _ptr(bias), _ptr(aux), _ptr(res),
              _ptr(out), _ptr(in_colsum), tag=tag, work=work)
        return
    _call("pm_gather_gemm_bf16", C.byref(desc), _ptr(inp), wsplit.data_ptr(), _ptr(bias), _ptr(aux), _ptr(res),
          _ptr(out), tag=tag, work=work)


USE_BF16_WGRAD = True   # weight gradients on the bf16 matrix cores when the shape qualifies


def _bf16_wgrad_channels(C: int) -> bool:
    """gathered channel counts pm_gather_wgrad_bf16 accepts: multiples of 32, or 32 < C < 64 (one zero-padded tap per
    k-block: the 48-channel layers of the VDVAE)"""
    return C % 32 == 0 or (32 < C < 64 and C % 4 == 0)


def _weight_extent(d: GatherDesc) -> int:
    """floats from a weight gradient's first element to one past the last the launch can write (all groups)"""
    one = ((d.KH - 1) * d.kws + d.KW - 1) * d.wts + (d.C - 1) * d.wcs + (d.N - 1) * d.wns + 1
    return (d.groups - 1) * d.w_gs + one


def _part_arenas(desc: GatherDesc, dw, db, nslots: int, dbg=None):
    """pm_wgrad_part of a launch whose weight (bias) gradients are ONE run of the flat gradient buffer starting at dw (db):
    plain launches and uniformly strided groups (the groups of every such launch in this package tile one contiguous run).
    None: dw is not a view of a parameter store's gradient buffer (free-standing tensors keep the atomics)."""
    from . import partials

    own = partials.owner_of(dw)
    if own is None or (db is not None and partials.owner_of(db) is not own) or (
            dbg is not None and partials.owner_of(dbg) is not own):
        return None
    wn = max(_weight_extent(desc), dw.numel())            # whole parameters (groups: the run their weights tile)
    wbuf, woff = own.arena(own.offset(dw), wn, nslots)
    wstride = own.entries[(own.offset(dw), wn)][2]
    bbuf = bgbuf = None
    boff = bgoff = bstride = bgstride = 0
    if db is not None:
        bn = max((desc.groups - 1) * desc.bias_gs + desc.N, db.numel())
        bbuf, boff = own.arena(own.offset(db), bn, nslots)
        bstride = own.entries[(own.offset(db), bn)][2]
    if dbg is not None:
        bgbuf, bgoff = own.arena(own.offset(dbg), dbg.numel(), nslots)
        bgstride = own.entries[(own.offset(dbg), dbg.numel())][2]
    part = partials.make_part(wbuf, woff, wstride, nslots, bbuf, boff, bstride, bgbuf, bgoff, bgstride)
    part._eager = (own, (own.offset(dw), wn), nslots * wstride * 4)     # see _part_done
    return part


def _part_done(part) -> None:
    """Behind a partial-sum launch: a LARGE arena (persistent-workgroup kernels: one slot per workgroup, 10 - 25 MB) is added
    into the flat buffer right away, on the launching stream, beside whatever the step's other stream runs - left to the
    reduction in front of the optimizer, its bytes would be read on the step's serial tail (measured: 181 MB = 52 us there
    for configs/pm_vae_mnist.py).  Small arenas (and every bias) stay for that one launch.  PM_PART_EAGER_MB: threshold."""
    own, key, nbytes = part._eager
    if nbytes >= _PART_EAGER_BYTES and key in own.pending and not torch.cuda.is_current_stream_capturing():
        own.reduce(key[0], key[0] + key[1], only=(key,))


# measured on configs/pm_vae_mnist.py (profiles/r04_ab_partials_vs_atomics.txt): reducing the 10 - 25 MB arenas behind their
# launches (12 extra launches) is SLOWER than one reduction in front of / inside the optimizer (1.385 vs 1.365 ms): off by default
_PART_EAGER_BYTES = float(os.environ.get("PM_PART_EAGER_MB", "1e9")) * (1 << 20)


def reduce_partials(table, njobs: int, flat_g, nbytes: float = 0.0) -> None:
    _call("pm_reduce_partials", table.data_ptr(), njobs, _ptr(flat_g), tag="reduce_partials_kernel", work={"bytes": nbytes})


def gather_wgrad(desc: GatherDesc, gathered, dense, dw, db, bf16: bool = False, db_gathered=None) -> None:
    """db_gathered: bias gradient taken over the GATHERED operand (transposed convs); only the thin lane form
    fuses it - callers must check the return value (True = db_gathered was accumulated).
    Gradients that are views of a ParamStore's flat buffer leave the kernels as PARTIAL SUMS (partials.PartialSums:
    plain stores into per-split slots, summed in a fixed order by pm_reduce_partials); free-standing dw / db keep the
    accumulate-with-atomics contract of the C entry points."""
    lib = _lib.load()
    if _lane_form(desc) and not os.environ.get("PM_NO_LANE_WGRAD"):
        tag = work = None
        if _timer is not None:
            tag = f"thin_wgrad_lane_kernel<{desc.C}, {desc.KH}, 32>"
            work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(gathered, dense, dw), "detail": _detail(desc)}
        ns = C.c_int()
        part = None
        if lib.pm_thin_wgrad_part_slots(C.byref(desc), _ptr(gathered), _ptr(dense), C.byref(ns)) == 0:
            part = _part_arenas(desc, dw, db, ns.value, db_gathered)
        if part is not None:
            _call("pm_thin_wgrad_part", C.byref(desc), _ptr(gathered), _ptr(dense), C.byref(part), tag=tag, work=work)
            _part_done(part)
            return True
        _call("pm_thin_wgrad", C.byref(desc), _ptr(gathered), _ptr(dense), _ptr(dw), _ptr(db), _ptr(db_gathered),
              tag=tag, work=work)
        return True
    al = int(gathered.data_ptr() % 16 == 0 and dense.data_ptr() % 16 == 0)
    if (bf16 and USE_BF16_WGRAD and _bf16_wgrad_channels(desc.C) and desc.N % 4 == 0 and desc.d in (1, 2)
            and gathered.data_ptr() % 16 == 0 and dense.data_ptr() % 16 == 0):
        tag = work = None
        if _timer is not None:
            rc, rn = 2 if desc.C % 64 == 0 and desc.KH * desc.KW * desc.C >= 64 else 1, 2 if desc.N > 32 else 1
            tag = (f"gather_wgrad_bf16_sub_kernel<{desc.d}>" if rc == 2 and rn == 2 else
                   f"gather_wgrad_bf16_kernel<{rc}, {rn}, {desc.d}>")
            if (_patch_form(desc) and desc.C == 32 and desc.N <= 64 and desc.KH * desc.KW in (9, 25)):
                tag = f"patch_wgrad_bf16_kernel<{2 if desc.N > 32 else 1}, {7 if desc.KH * desc.KW == 25 else 3}>"
            work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(gathered, dense, dw), "detail": _detail(desc)}
        ns = C.c_int()
        part = None
        if lib.pm_wgrad_part_slots(C.byref(desc), _ptr(gathered), _ptr(dense), 0, al, 1, C.byref(ns)) == 0:
            part = _part_arenas(desc, dw, db, ns.value)
        if part is not None:
            _call("pm_gather_wgrad_part", C.byref(desc), _ptr(gathered), _ptr(dense), None, al, 1, C.byref(part), tag=tag,
                  work=work)
            _part_done(part)
            return
        _call("pm_gather_wgrad_bf16", C.byref(desc), _ptr(gathered), _ptr(dense), _ptr(dw), _ptr(db), tag=tag, work=work)
        return
    tag = work = None
    if _timer is not None:
        v = [C.c_int() for _ in range(5)]
        lib.pm_query_wgrad_plan(C.byref(desc), int(gathered.data_ptr() % 16 == 0),
                                int(dense.data_ptr() % 16 == 0), *[C.byref(x) for x in v])
        tag = f"gather_wgrad_kernel<{v[0].value},{v[1].value},{_MODES[v[2].value]},{v[3].value}>"
        work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(gathered, dense, dw), "detail": _detail(desc)}
    ns = C.c_int()
    part = None
    if lib.pm_wgrad_part_slots(C.byref(desc), _ptr(gathered), _ptr(dense), 0, al, 0, C.byref(ns)) == 0:
        part = _part_arenas(desc, dw, db, ns.value)
    if part is not None:
        _call("pm_gather_wgrad_part", C.byref(desc), _ptr(gathered), _ptr(dense), None, al, 0, C.byref(part), tag=tag, work=work)
        _part_done(part)
        return
    _call("pm_gather_wgrad", C.byref(desc), _ptr(gathered), _ptr(dense), _ptr(dw), _ptr(db), tag=tag, work=work)


def _lane_form(d: GatherDesc) -> bool:
    """csrc/pm_thin.hip lane_form_ok: stride-1 layers whose thin side has 1 or 2 channels"""
    return (d.groups == 1 and d.d == 1 and d.a == 1 and d.KH == d.KW and d.KH in (3, 5) and d.C in (1, 2)
            and d.N <= 32 and d.OW <= 128 and d.OH <= 128 and d.off_x == d.off and d.kws == d.KW)


def _to1_form(d: GatherDesc) -> bool:
    """wide -> 1 channel, stride 1 (thin_to1_kernel)"""
    return (d.groups == 1 and d.d == 1 and d.a == 1 and d.N == 1 and d.C % 4 == 0 and d.OW <= 256
            and d.off_x == d.off and d.kws == d.KW
            and (d.KH * d.KW * d.C + d.KH * (d.OW + d.KW - 1) * (d.C + 4)) * 4 <= 64 * 1024)


def _thin_ok(d: GatherDesc) -> bool:
    if _lane_form(d) or _to1_form(d):
        return True
    return (d.groups == 1 and d.d == 1 and d.KH * d.KW * d.C <= 64 and d.N <= 32 and d.N % 8 == 0
            and d.off_x == d.off and d.kws == d.KW)


def thin_conv(desc: GatherDesc, inp, w, bias, aux, res, out) -> None:
    tag = work = None
    if _timer is not None:
        tag = ("thin_to1_kernel" if _to1_form(desc) and inp.data_ptr() % 16 == 0 else
               f"thin_conv_lane_kernel<{desc.C}, {desc.KH}>" if _lane_form(desc) else "thin_conv_kernel")
        work = {"flops": _algorithmic_flops(desc), "bytes": _nbytes(inp, aux, res, out), "detail": _detail(desc)}
    _call("pm_thin_conv", C.byref(desc), _ptr(inp), _ptr(w), _ptr(bias), _ptr(aux), _ptr(res), _ptr(out), tag=tag,
          work=work)


def layer_forward(g: LayerGeom, x, w, b, out, in_act=ACT_NONE, out_act=ACT_NONE, res=None, wsplit=None,
                  tmp=None, out2=None, act2=ACT_NONE, **group_kw) -> None:
    """wsplit: this layer's pre-split bf16 weights for the forward direction (ParamStore.split_view);
    when given and the shape qualifies the layer runs on the bf16 matrix cores (bf16x3).
    tmp: [B, IH, IW, k*k] scratch for a wide -> 1-channel transposed conv (per-tap dot products)."""
    B = group_kw.pop("B", None) or x.shape[0]
    d = g._desc(B, "fwd", **group_kw)
    d.in_act, d.out_act = in_act, out_act
    if (wsplit is not None and _to1_form(d) and d.C in (32, 64) and d.KH * d.KW <= 32 and x.data_ptr() % 16 == 0
            and d.KH * d.IW * ((d.KH * d.KW) | 1) * 4 <= 64 * 1024):
        work = {"flops": _algorithmic_flops(d), "bytes": _nbytes(x, res, out), "detail": _detail(d)}
        _call("pm_thin_to1_bf16", C.byref(d), _ptr(x), _ptr(w), _ptr(b), None, _ptr(res), _ptr(out),
              tag=f"thin_to1_bf16_kernel<{d.C // 32}>", work=work)
        if out2 is not None:
            gelu_fwd(out, None, out2)
        return
    if _thin_ok(d):
        thin_conv(d, x, w, b, None, res, out)
        if out2 is not None:
            gelu_fwd(out, None, out2)
        return
    if g.kind == "convT" and g.CO == 1 and g.s == 1 and tmp is not None and res is None and in_act == ACT_NONE:
        # out[p] = sum_tap (x[p + tap] . w[tap]):  T = x @ W' (one GEMM, N = taps) then a shifted sum
        taps = g.k * g.k
        dt = LayerGeom.dense(g.CI, taps)._desc(B * g.IH * g.IW, "fwd")
        dt.wts, dt.wcs, dt.wns = 0, 1, g.CI               # W'[c][tap] = w[tap, 0, c]
        gather_gemm(dt, x, w, None, None, None, tmp)
        work = {"bytes": _nbytes(tmp, out), "detail": _detail(d)}
        _call("pm_tap_shift_add", C.byref(d), _ptr(tmp), taps, _ptr(b), _ptr(out), work=work)
        return
    if wsplit is not None and bf16_supported(d):
        if d.groups > 1:
            d.w_gs = wsplit.numel() // d.groups
        gather_gemm_bf16(d, x, wsplit, b, None, res, out, out2=out2, act2=act2)
    else:
        gather_gemm(d, x, w, b, None, res, out)
        if out2 is not None:
            assert act2 == ACT_GELU, "the unfused second store only exists for gelu"
            gelu_fwd(out, None, out2)


def dgrad_insum_ok(g: LayerGeom, B: int, wsplit, force: bool = False) -> bool:
    """True when layer_dgrad(g, dy, ..., in_colsum=db) can also produce db = column sums of dy (the bias gradient of a
    transposed convolution) - its dy images are staged in LDS by the image-resident kernel anyway.  Measured on the PM-VAE step
    (same box, 300 steps each): 189.8 / 191.6 k img/s with it against 193.8 / 192.7 k with the separate pm_colsum launches - the
    column sums ride on the weight-gradient stream, the data gradient is the critical chain, and five launches fewer do not pay
    for the atomics and the longer staging loop there.  So the models only use it when PM_DGRAD_INSUM=1 (`force`: tests) - and
    the default libpmhip.so is built WITHOUT the code (pm_image_conv_insum_applies returns 0; see csrc/pm_conv.hip)."""
    if wsplit is None or not (force or os.environ.get("PM_DGRAD_INSUM")):
        return False
    d = g._desc(B, "dgrad")
    if _thin_ok(d) or not bf16_supported(d):
        return False
    return bool(_lib.load().pm_image_conv_insum_applies(C.byref(d)))


def layer_dgrad(g: LayerGeom, dy, w, dx, aux=None, aux_act=ACT_NONE, res=None, wsplit=None, in_colsum=None, **group_kw) -> None:
    B = group_kw.pop("B", None) or dy.shape[0]
    d = g._desc(B, "dgrad", **group_kw)
    d.aux_act = aux_act if aux is not None else ACT_NONE
    aux = aux if aux_act != ACT_NONE else None
    if in_colsum is not None:                       # the caller asked dgrad_insum_ok first
        gather_gemm_bf16(d, dy, wsplit, None, aux, res, dx, in_colsum=in_colsum)
        return
    if _thin_ok(d):
        thin_conv(d, dy, w, None, aux, res, dx)
        return
    if wsplit is not None and bf16_supported(d):
        if d.groups > 1:
            d.w_gs = wsplit.numel() // d.groups
        gather_gemm_bf16(d, dy, wsplit, None, aux, res, dx)
    else:
        gather_gemm(d, dy, w, None, aux, res, dx)


def layer_wgrad(g: LayerGeom, x, dy, dw, db, in_act=ACT_NONE, bf16: bool = True, **group_kw) -> None:
    B = group_kw.pop("B", None) or x.shape[0]
    d = g._desc(B, "wgrad", **group_kw)
    d.in_act = in_act
    if g.kind != "convT":
        # dw[ky,kx,ci,co] = sum_{b,p,q} x[b, p*s+ky-pad, q*s+kx-pad, ci] * dy[b,p,q,co]
        gather_wgrad(d, x, dy, dw, db, bf16=bf16)
        return
    # Transposed conv: dw[ky,kx,co,ci] = sum_{b,oy,ox} dy[b,oy,ox,co] * xdil[b,oy+ky-pa,ox+kx-pa,ci].
    # Walking the OUTPUT grid multiplies the zeros of the dilated x (3 of 4 rows at stride 2, 48 of
    # 49 for the 1x1 -> 7x7 layer).  Walk the INPUT grid instead with the roles swapped:
    #   dw[ky,kx,co,ci] = sum_{b,iy,ix} dy[b, iy*s+pa-ky, ix*s+pa-kx, co] * x[b,iy,ix,ci]
    # which is this layer's data-gradient index rule with dy gathered and x dense.
    if in_act != ACT_NONE:
        raise NotImplementedError("transposed-conv weight gradient with a pending input activation")
    d = g._desc(B, "dgrad", **group_kw)
    fused = gather_wgrad(d, dy, x, dw, None, bf16=bf16, db_gathered=db)
    if db is not None and not fused:
        colsum(dy, db)


class WgradBatch:
    """Weight gradients collected during a backward pass and launched at its end, ONE launch per geometry
    (pm_gather_wgrad_table): the weight / bias gradients of a layer only feed the optimizer, and a model made of many
    same-shaped small layers (the VDVAE's 100 bottleneck Blocks: 400 weight-gradient launches of ~25 us, a third of its
    step) pays one launch per (resolution, layer shape) instead.  Every operand is a static workspace / parameter buffer,
    so the per-geometry offset tables are built once and a recorded launch plan replays the same launches."""

    def __init__(self):
        self.items = {}          # key -> [(geom, x, dy, dw, db)]
        self._tables = {}        # (key, pointer tuple) -> (device table, aligned flag)
        self._ptables = {}       # (key, pointer tuple) -> partial-sum form of the same launch (_part_table)
        self.reducer = None      # parallel.GradReducer under data parallelism: every grouped launch reports its weight ranges

    def add(self, g: LayerGeom, x, dy, dw, db, bf16: bool, in_act: int = ACT_NONE) -> None:
        if g.kind == "convT":
            raise NotImplementedError("deferred weight gradients of transposed convolutions")
        B = x.shape[0]
        d = g._desc(B, "wgrad")
        use_bf16 = bool(bf16 and USE_BF16_WGRAD and _bf16_wgrad_channels(d.C) and d.N % 4 == 0 and d.d in (1, 2))
        key = (B, g.kind, g.IH, g.IW, g.CI, g.OH, g.OW, g.CO, g.KH, g.KW, g.s, g.pad, g.pad_x, g.full_kh, g.full_kw, in_act,
               use_bf16, db is not None)
        self.items.setdefault(key, []).append((g, x, dy, dw, db))

    @staticmethod
    def eligible(g: LayerGeom, d: GatherDesc) -> bool:
        """shapes the grouped kernels serve: the MFMA weight-gradient forms (not the 1- / 2-channel lane kernels)"""
        return g.kind != "convT" and not _lane_form(d)

    def flush(self) -> None:
        for key, lst in self.items.items():
            g0, x0, dy0, dw0, db0 = lst[0]
            ptrs = tuple(t.data_ptr() for it in lst for t in it[1:] if t is not None)
            cached = self._tables.get((key, ptrs))
            if cached is None:
                rows = []
                for _, x, dy, dw, db in lst:
                    rows += [(x.data_ptr() - x0.data_ptr()) // 4, (dy.data_ptr() - dy0.data_ptr()) // 4,
                             (dw.data_ptr() - dw0.data_ptr()) // 4,
                             ((db.data_ptr() - db0.data_ptr()) // 4) if db is not None else 0]
                aligned = all(x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 for _, x, dy, _, _ in lst)
                cached = (torch.tensor(rows, dtype=torch.int64, device=x0.device), int(aligned))
                self._tables[(key, ptrs)] = cached
            table, aligned = cached
            d = g0._desc(key[0], "wgrad", groups=len(lst))
            d.in_act = key[-3]
            tag = work = None
            if _timer is not None:
                tag = "gather_wgrad_table"
                work = {"flops": _algorithmic_flops(d), "bytes": len(lst) * _nbytes(x0, dy0, dw0), "detail": _detail(d)}
            pt = self._part_table(key, ptrs, lst, d, aligned)
            if pt is not None:          # partial sums: one slot-major arena for the launch, [slot][group][weight]
                ptable, part = pt
                _call("pm_gather_wgrad_part", C.byref(d), _ptr(x0), _ptr(dy0), ptable.data_ptr(), aligned, int(key[-2]),
                      C.byref(part), tag=tag, work=work)
            else:
                _call("pm_gather_wgrad_table", C.byref(d), _ptr(x0), _ptr(dy0), _ptr(dw0), _ptr(db0), table.data_ptr(), aligned,
                      int(key[-2]), tag=tag, work=work)
            if self.reducer is not None:
                # data-parallel: these weights have their final gradient once this launch has run (a parameter belongs to one
                # layer, a layer to one group); bias gradients sit in the 1-D suffix, which finish() reduces
                base, n = self.reducer.flat.data_ptr(), self.reducer.flat.numel()
                offs = [((dw.data_ptr() - base) // 4, dw.numel()) for _, _, _, dw, _ in lst]
                self.reducer.ready_ranges([(o, o + c) for o, c in offs if 0 <= o and o + c <= n])
        self.items = {}

    def _part_table(self, key, ptrs, lst, d, aligned):
        """(device table with arena-relative dw / db offsets, pm_wgrad_part) when every gradient of the launch is a view of
        one ParamStore's flat buffer; None keeps the accumulate-with-atomics launch"""
        from . import partials

        g0, x0, dy0, dw0, db0 = lst[0]
        own = partials.owner_of(dw0)
        if own is None or any(partials.owner_of(it[3]) is not own or (it[4] is not None and partials.owner_of(it[4]) is not own)
                              for it in lst):
            return None
        cached = self._ptables.get((key, ptrs))
        if cached is None:
            ns = C.c_int()
            if _lib.load().pm_wgrad_part_slots(C.byref(d), _ptr(x0), _ptr(dy0), 1, aligned, int(key[-2]), C.byref(ns)) != 0:
                return None
            S, G = ns.value, len(lst)
            wn = (dw0.numel() + 3) // 4 * 4
            bn = ((db0.numel() + 3) // 4 * 4) if db0 is not None else 0
            wbuf = own.new_shared(S * G * wn)
            bbuf = own.new_shared(S * G * bn) if bn else None
            rows = []
            for gi, (_, x, dy, dw, db) in enumerate(lst):
                rows += [(x.data_ptr() - x0.data_ptr()) // 4, (dy.data_ptr() - dy0.data_ptr()) // 4, gi * wn, gi * bn]
            ptable = torch.tensor(rows, dtype=torch.int64, device=x0.device)
            part = partials.make_part(wbuf, 0, G * wn, S, bbuf, 0, G * bn)
            cached = (ptable, part, wbuf, bbuf, S, wn, bn)
            self._ptables[(key, ptrs)] = cached
        ptable, part, wbuf, bbuf, S, wn, bn = cached
        G = len(lst)
        for gi, (_, _, _, dw, db) in enumerate(lst):       # (re-)register the runs and mark them pending
            own.arena(own.offset(dw), dw.numel(), S, stride=G * wn, shared=(wbuf, gi * wn))
            if db is not None:
                own.arena(own.offset(db), db.numel(), S, stride=G * bn, shared=(bbuf, gi * bn))
        return ptable, part

    def discard(self) -> None:
        """drops queued items (a backward pass that raised must not leave stale operands behind)"""
        self.items = {}


# ------------------------------------------------------------------------------------------
# heads / loss / optimizer
# ------------------------------------------------------------------------------------------
def mask_concat(x, b, out) -> None:
    C_, Cb = x.shape[-1], b.shape[-1]
    R = x.numel() // C_
    _call("pm_mask_concat", _ptr(x), _ptr(b), _ptr(out), R, C_, Cb)


def tril_sample_kl_fwd(params, eps, z, kl) -> None:
    B, k = eps.shape
    _call("pm_tril_sample_kl_fwd", _ptr(params), _ptr(eps), _ptr(z), _ptr(kl), B, k)


def tril_sample_kl_bwd(params, eps, dz, g_kl, dparams, dz2=None) -> None:
    """dz2: a second gradient w.r.t. z, added as the rows are loaded (no separate axpy launch)"""
    B, k = eps.shape
    if dz2 is not None:
        _call("pm_tril_sample_kl_bwd2", _ptr(params), _ptr(eps), _ptr(dz), _ptr(dz2), _ptr(g_kl), _ptr(dparams), B, k,
              tag="pm_tril_sample_kl_bwd")
        return
    _call("pm_tril_sample_kl_bwd", _ptr(params), _ptr(eps), _ptr(dz), _ptr(g_kl), _ptr(dparams), B, k)


def tril_logprob_fwd(params, z, lp) -> None:
    B, k = z.shape
    _call("pm_tril_logprob_fwd", _ptr(params), _ptr(z), _ptr(lp), B, k)


def tril_logprob_bwd(params, z, g, dparams, dz) -> None:
    B, k = z.shape
    _call("pm_tril_logprob_bwd", _ptr(params), _ptr(z), _ptr(g), _ptr(dparams), _ptr(dz), B, k)


def diag_gaussian_sample_kl_fwd(params, eps, z, kl) -> None:
    B, k = eps.shape
    _call("pm_diag_gaussian_sample_kl_fwd", _ptr(params), _ptr(eps), _ptr(z), _ptr(kl), B, k)


def diag_gaussian_sample_kl_bwd(params, eps, dz, g_kl, dparams) -> None:
    B, k = eps.shape
    _call("pm_diag_gaussian_sample_kl_bwd", _ptr(params), _ptr(eps), _ptr(dz), _ptr(g_kl), _ptr(dparams), B, k)


def diag_gaussian_logprob_fwd(params, z, lp) -> None:
    B, k = z.shape
    _call("pm_diag_gaussian_logprob_fwd", _ptr(params), _ptr(z), _ptr(lp), B, k)


def diag_gaussian_logprob_bwd(params, z, g, dparams, dz) -> None:
    B, k = z.shape
    _call("pm_diag_gaussian_logprob_bwd", _ptr(params), _ptr(z), _ptr(g), _ptr(dparams), _ptr(dz), B, k)


def bernoulli_ll_fwd(logits, x, ll) -> None:
    B = x.shape[0]
    _call("pm_bernoulli_ll_fwd", _ptr(logits), _ptr(x), _ptr(ll), B, x.numel() // B)


def bernoulli_ll_fwd_bwd(logits, x, g, ll, dpre, act, slope=LEAKY_SLOPE) -> None:
    """ll [B] and d loss / d pre-activation in one pass (g [B] = d loss / d ll, known before the forward pass)"""
    B = x.shape[0]
    _call("pm_bernoulli_ll_fwd_bwd", _ptr(logits), _ptr(x), _ptr(g), _ptr(ll), _ptr(dpre), B, x.numel() // B, act, slope)


def bernoulli_ll_bwd(logits, x, g, dpre, act, slope=LEAKY_SLOPE) -> None:
    B = x.shape[0]
    _call("pm_bernoulli_ll_bwd", _ptr(logits), _ptr(x), _ptr(g), _ptr(dpre), B, x.numel() // B, act, slope)


def normal_ll_fwd(loc, x, log_scale, ll, scale_eps: float = 0.0) -> None:
    B = x.shape[0]
    _call("pm_normal_ll_fwd", _ptr(loc), _ptr(x), _ptr(log_scale), _ptr(ll), B, x.numel() // B, scale_eps)


_NLL_SCRATCH: dict = {}     # (device, stream, B) -> [B + 1] floats: example terms + the ticket of pm_normal_ll_bwd_det


def normal_ll_bwd(loc, x, log_scale, g, dloc, d_log_scale, scale_eps: float = 0.0) -> None:
    B = x.shape[0]
    if not os.environ.get("PM_NLL_ATOMIC"):                 # A/B knob: the f32-atomic d_log_scale of rounds 1-3
        key = (x.device.index, _stream(), B)               # launches on one stream run in order: one scratch per stream
        scratch = _NLL_SCRATCH.get(key)
        if scratch is None:
            scratch = _NLL_SCRATCH[key] = torch.zeros(B + 1, dtype=torch.float32, device=x.device)
        _call("pm_normal_ll_bwd_det", _ptr(loc), _ptr(x), _ptr(log_scale), _ptr(g), _ptr(dloc), _ptr(d_log_scale), B,
              x.numel() // B, scale_eps, scratch.data_ptr())
        return
    _call("pm_normal_ll_bwd", _ptr(loc), _ptr(x), _ptr(log_scale), _ptr(g), _ptr(dloc), _ptr(d_log_scale), B,
          x.numel() // B, scale_eps)


# ---- vector quantisation (hk.nets.VectorQuantizerEMA, reference vqvae.py:66-72,80) -----------------
_VQ_SCRATCH: dict = {}      # (device, stream, floats) -> segment sums of pm_vq_dw_exact


def vq_select(z, emb, dots, e2, idx, quant, commit_grad, sqerr, counts, dw, commit_coef: float) -> None:
    """z [N,D], emb [D,K], dots [N,K] = z @ emb."""
    D, K = emb.shape
    N = z.numel() // D
    work = {"bytes": _nbytes(z, dots, quant, commit_grad), "detail": f"N{N} D{D} K{K}"}
    if dw is not None and D <= 1024 and not os.environ.get("PM_VQ_DW_ATOMIC"):     # A/B knob: dw from float atomics
        _call("pm_vq_select", _ptr(z), _ptr(emb), _ptr(dots), _ptr(e2), _iptr(idx), _ptr(quant), _ptr(commit_grad),
              _ptr(sqerr), _ptr(counts), None, N, D, K, commit_coef, work=work)
        n = C.c_longlong(0)
        _lib.load().pm_vq_dw_exact_floats(N, D, K, C.byref(n))
        key = (z.device.index, _stream(), n.value)
        scratch = _VQ_SCRATCH.get(key)
        if scratch is None and n.value:
            scratch = _VQ_SCRATCH[key] = torch.empty(n.value, dtype=torch.float32, device=z.device)
        _call("pm_vq_dw_exact", _ptr(z), _iptr(idx), _ptr(dw), N, D, K, _ptr(scratch), n.value,
              work={"bytes": _nbytes(z, dw) + 4 * N, "detail": f"N{N} D{D} K{K}"})
        return
    _call("pm_vq_select", _ptr(z), _ptr(emb), _ptr(dots), _ptr(e2), _iptr(idx), _ptr(quant), _ptr(commit_grad),
          _ptr(sqerr), _ptr(counts), _ptr(dw), N, D, K, commit_coef, work=work)


def vq_ema_update(counts, dw, cs_hidden, cs_avg, dw_hidden, dw_avg, emb, counter, decay: float, epsilon: float) -> None:
    D, K = emb.shape
    _call("pm_vq_ema_update", _ptr(counts), _ptr(dw), _ptr(cs_hidden), _ptr(cs_avg), _ptr(dw_hidden), _ptr(dw_avg),
          _ptr(emb), _iptr(counter), D, K, decay, epsilon)


def vq_lookup(idx, emb, quant) -> None:
    D, K = emb.shape
    _call("pm_vq_lookup", _iptr(idx), _ptr(emb), _ptr(quant), idx.numel(), D, K)


def vqvae_loss(ll, sqerr, counts, D: int, commitment_cost: float, grad_scale: float, out, g_ll) -> None:
    _call("pm_vqvae_loss", _ptr(ll), _ptr(sqerr), _ptr(counts), ll.numel(), sqerr.numel(), D, counts.numel(),
          commitment_cost, grad_scale, _ptr(out), _ptr(g_ll))


def argmm_build_input(z, ctx, inp) -> None:
    B, k = z.shape
    _call("pm_argmm_build_input", _ptr(z), _ptr(ctx), _ptr(inp), B, k, ctx.numel() // B)


def argmm_input_bwd(dinp, dz, dctx, B, k, ctx_dim, accumulate_dz, ctx=None, ctx_act=ACT_NONE) -> None:
    _call("pm_argmm_input_bwd", _ptr(dinp), _ptr(dz), _ptr(dctx), B, k, ctx_dim, int(accumulate_dz), _ptr(ctx), ctx_act, LEAKY_SLOPE)


def gmm_logprob_fwd(head, z, mll, nc) -> None:
    B, k = z.shape
    _call("pm_gmm_logprob_fwd", _ptr(head), _ptr(z), _ptr(mll), B, k, nc)


def gmm_logprob_bwd(head, z, g, dhead, dz, nc, accumulate_dz) -> None:
    B, k = z.shape
    _call("pm_gmm_logprob_bwd", _ptr(head), _ptr(z), _ptr(g), _ptr(dhead), _ptr(dz), B, k, nc, int(accumulate_dz))


# ---- PixelCNN row-wise pieces (reference pixel_cnn.py:372-553) ----------------------------------------
def embed_fwd(idx, table, out) -> None:
    K, F = table.shape
    _call("pm_embed_fwd", _iptr(idx), _ptr(table), _ptr(out), idx.numel(), F, K)


_embed_scratch = {}


def embed_bwd(idx, dout, dtable) -> None:
    """dtable[idx[r]] += dout[r] in a fixed order (pm_embed_bwd_sorted; F > 1024 or PM_EMBED_FIXEDPOINT=1: 64-bit fixed-point
    integer atomics, pm_embed_bwd_exact) - the same bits in every run; PM_EMBED_ATOMIC=1: the f32-atomic form (A/B)"""
    K, F = dtable.shape
    if os.environ.get("PM_EMBED_ATOMIC"):
        _call("pm_embed_bwd", _iptr(idx), _ptr(dout), _ptr(dtable), idx.numel(), F, K)
        return
    if F <= 1024 and not os.environ.get("PM_EMBED_FIXEDPOINT"):        # A/B knob: the fixed-point atomic form
        n = -(-idx.numel() // 512) * K * F             # room for the shortest segments (pm_embed_bwd_sorted picks)
        key = (dtable.device.index, _stream(), n)
        scratch = _VQ_SCRATCH.get(key)
        if scratch is None:
            scratch = _VQ_SCRATCH[key] = torch.empty(n, dtype=torch.float32, device=dtable.device)
        _call("pm_embed_bwd_sorted", _iptr(idx), _ptr(dout), _ptr(dtable), idx.numel(), F, K, scratch.data_ptr(), n)
        return
    key = (dtable.data_ptr(), K, F)
    scratch = _embed_scratch.get(key)
    if scratch is None:
        if len(_embed_scratch) > 64:
            _embed_scratch.clear()
        scratch = _embed_scratch[key] = torch.zeros(K * F + 1, dtype=torch.int64, device=dtable.device)
    _call("pm_embed_bwd_exact", _iptr(idx), _ptr(dout), _ptr(dtable), idx.numel(), F, K, scratch.data_ptr())


@dataclass
class PhiloxDrop:
    """hk.dropout keep mask that is never materialised: concat_elu_fwd / _bwd draw it in place from the Philox stream
    (seed, step counter on the device, stream_id) - the values dropout_mask() would write with the same arguments"""

    rate: float
    seed: int
    step_dev: "torch.Tensor"
    stream_id: int

    @staticmethod
    def usable(*tensors) -> bool:
        """the in-place form is the 16-byte kernel: channel counts % 4 == 0, 16-byte aligned operands"""
        return all(t is None or (t.shape[-1] % 4 == 0 and t.data_ptr() % 16 == 0) for t in tensors)


def concat_elu_fwd(a, b, drop, out) -> None:
    """drop: None, a pre-scaled keep-mask tensor shaped like `out`, or a PhiloxDrop (mask drawn in place)"""
    Ca, Cb = a.shape[-1], (b.shape[-1] if b is not None else 0)
    if isinstance(drop, PhiloxDrop):
        _call("pm_concat_elu_fwd_philox", _ptr(a), _ptr(b), _ptr(out), a.numel() // Ca, Ca, Cb, drop.rate,
              drop.seed & (2 ** 64 - 1), _iptr(drop.step_dev), drop.stream_id, work={"bytes": _nbytes(a, b, out)})
        return
    _call("pm_concat_elu_fwd", _ptr(a), _ptr(b), _ptr(drop), _ptr(out), a.numel() // Ca, Ca, Cb,
          work={"bytes": _nbytes(a, b, drop, out)})


def concat_elu_bwd(a, b, drop, dout, da, db, accumulate: bool, add_a=None) -> None:
    """add_a (same shape as a): added to da in the same pass (a gated block's residual path d_in += dout)"""
    Ca, Cb = a.shape[-1], (b.shape[-1] if b is not None else 0)
    if isinstance(drop, PhiloxDrop):
        _call("pm_concat_elu_bwd_philox", _ptr(a), _ptr(b), _ptr(dout), _ptr(da), _ptr(db), a.numel() // Ca, Ca, Cb,
              int(accumulate), _ptr(add_a), drop.rate, drop.seed & (2 ** 64 - 1), _iptr(drop.step_dev), drop.stream_id,
              work={"bytes": _nbytes(a, b, dout, da, db, add_a)})
        return
    _call("pm_concat_elu_bwd", _ptr(a), _ptr(b), _ptr(drop), _ptr(dout), _ptr(da), _ptr(db), a.numel() // Ca, Ca, Cb,
          int(accumulate), _ptr(add_a), work={"bytes": _nbytes(a, b, drop, dout, da, db, add_a)})


def gate_fwd(y, h, inp, out, P: int) -> None:
    F = inp.shape[-1]
    _call("pm_gate_fwd", _ptr(y), _ptr(h), _ptr(inp), _ptr(out), inp.numel() // F, F, P, work={"bytes": _nbytes(y, inp, out)})


def gate_fwd_ce_ok(y, h, inp, out, ce) -> bool:
    F = inp.shape[-1]
    return (F % 4 == 0 and inp.numel() * 2 < 0x7fffffff
            and all(t is None or t.data_ptr() % 16 == 0 for t in (y, h, inp, out, ce)))


def gate_fwd_ce(y, h, inp, out, ce, P: int) -> None:
    """gate_fwd + concat_elu_fwd(out) -> ce [.., 2F] in one launch (gate_fwd_ce_ok)"""
    F = inp.shape[-1]
    _call("pm_gate_fwd_ce", _ptr(y), _ptr(h), _ptr(inp), _ptr(out), _ptr(ce), inp.numel() // F, F, P,
          work={"bytes": _nbytes(y, inp, out, ce)})


def gate_bwd_rows_sum_ok(dout, B: int) -> bool:
    """pm_gate_bwd_rows_sum's preconditions, and enough images to fill the chip with one workgroup each"""
    F = dout.shape[-1]
    return F % 4 == 0 and F // 4 <= 64 and 256 % (F // 4) == 0 and B >= 128 and dout.data_ptr() % 16 == 0


def gate_bwd_rows_sum(y, h, dout, dy, dh, P: int) -> None:
    """gate_bwd + rows_sum(dy) in one launch: dh [B, 2F] = sum over the P positions of dy"""
    F = dout.shape[-1]
    _call("pm_gate_bwd_rows_sum", _ptr(y), _ptr(h), _ptr(dout), _ptr(dy), _ptr(dh), dout.numel() // F // P, F, P,
          work={"bytes": _nbytes(y, dout, dy)})


def gate_bwd(y, h, dout, dy, P: int) -> None:
    F = dout.shape[-1]
    _call("pm_gate_bwd", _ptr(y), _ptr(h), _ptr(dout), _ptr(dy), dout.numel() // F, F, P, work={"bytes": _nbytes(y, dout, dy)})


def rows_sum(x, out, P: int) -> None:
    N = x.shape[-1]
    _call("pm_rows_sum", _ptr(x), _ptr(out), x.numel() // (N * P), N, P, work={"bytes": _nbytes(x, out)})


ROWS_SUM_MULTI_MAX = 64


def rows_sum_multi(xs, out, P: int) -> None:
    """out[g] = rows_sum(xs[g]) for <= 64 tensors of one shape in one launch (pm_rows_sum_multi)"""
    N = xs[0].shape[-1]
    arr = _ptr_array(xs)                    # host array of device pointers: lives in the recorded call's argument tuple
    _call("pm_rows_sum_multi", arr, len(xs), _ptr(out), xs[0].numel() // (N * P), N, P,
          work={"bytes": sum(_nbytes(x) for x in xs) + _nbytes(out)})


def groups_sum(x, out, G: int, accumulate: bool = False) -> None:
    _call("pm_groups_sum", _ptr(x), _ptr(out), out.numel(), G, x.numel() // G, int(accumulate))


def elu_fwd(x, out) -> None:
    _call("pm_elu_fwd", _ptr(x), _ptr(out), x.numel())


def elu_bwd(x, dout, dx, accumulate: bool = False) -> None:
    _call("pm_elu_bwd", _ptr(x), _ptr(dout), _ptr(dx), x.numel(), int(accumulate))


def categorical_ll_fwd(logits, idx, lse, ll, P: int) -> None:
    K = logits.shape[-1]
    _call("pm_categorical_ll_fwd", _ptr(logits), _iptr(idx), _ptr(lse), _ptr(ll), logits.numel() // K, K, P)


def categorical_ll_bwd(logits, idx, lse, g, dlogits, P: int) -> None:
    K = logits.shape[-1]
    _call("pm_categorical_ll_bwd", _ptr(logits), _iptr(idx), _ptr(lse), _ptr(g), _ptr(dlogits), logits.numel() // K, K, P)


def neg_mean_loss(ll, grad_scale: float, out, g_ll) -> None:
    _call("pm_neg_mean_loss", _ptr(ll), ll.numel(), grad_scale, _ptr(out), _ptr(g_ll))


def categorical_sample(logits, gumbel, idx, P: int, pos: int) -> None:
    """logits [B*P, K], gumbel [B, K], idx int32 [B*P] updated at position `pos` of every example"""
    K = logits.shape[-1]
    _call("pm_categorical_sample", _ptr(logits), _ptr(gumbel), _iptr(idx), gumbel.shape[0], K, P, pos)


def impute_blend(x, mask, imp, lo: float = 0.0, hi: float = 1.0) -> None:
    """imp [B,S,...image] <- clip(where(mask, x, imp), lo, hi) in place (lo > hi: no clipping)"""
    B, S = imp.shape[0], imp.shape[1]
    D, C, Cm = x.numel() // B, x.shape[-1], mask.shape[-1]
    _call("pm_impute_blend", _ptr(x), _ptr(mask), _ptr(imp), B, S, D, C, Cm, lo, hi)


def imputation_psnr(imp, x, psnr, scale: float = 1.0) -> None:
    B, S = imp.shape[0], imp.shape[1]
    _call("pm_imputation_psnr", _ptr(imp), _ptr(x), _ptr(psnr), B, S, x.numel() // B, scale)


def mlp_pair_bf16(x, w1_split, w2_split, b1, b2, aux1, aux2, out1, out2, in_act, mid_act, aux_act1, aux_act2) -> None:
    """csrc/pm_mlp.hip: two 256 -> 256 dense layers of a ResidualMLP block in one launch (see include/pmhip.h)"""
    R, hid = x.shape[0], x.shape[1]
    work = {"flops": 4.0 * R * hid * hid, "bytes": _nbytes(x, aux1, aux2, out1, out2), "detail": f"B{R} 2x({hid}->{hid})"}
    _call("pm_mlp_pair_bf16", _ptr(x), w1_split.data_ptr(), w2_split.data_ptr(), _ptr(b1), _ptr(b2), _ptr(aux1), _ptr(aux2),
          _ptr(out1), _ptr(out2), R, hid, in_act, mid_act, aux_act1, aux_act2, LEAKY_SLOPE, tag="mlp_pair_bf16_kernel",
          work=work)


def info_gain_inputs(x, b, x_u, out) -> None:
    """csrc/pm_eval.hip: the [F + 1] candidate rows concat([x * m_c, m_c]) of one decoder sample (vae.py:257-276)"""
    C_, Cb = x.shape[-1], b.shape[-1]
    P = x.numel() // C_
    _call("pm_info_gain_inputs", _ptr(x), _ptr(b), _ptr(x_u), _ptr(out), P, C_, Cb)


def gaussian_entropy(params, ent, k: int, tril: bool) -> None:
    """entropy per row of a TriL / diagonal Gaussian head; `ent` may be a strided view (one column of a matrix)"""
    assert ent.is_cuda and ent.dtype == torch.float32
    _call("pm_gaussian_entropy", _ptr(params), ent.data_ptr(), params.shape[0], k, int(tril), ent.stride(0) if ent.dim() else 1)


def info_gain_finish(ents, b, gains) -> None:
    S, F1 = ents.shape
    _call("pm_info_gain_finish", _ptr(ents), _ptr(b), _ptr(gains), S, F1 - 1)


def mlp_chain_ok(rows: int, hidden: int) -> bool:
    """preconditions of pm_mlp_chain_bf16: hidden width 256, whole 64-row tiles"""
    return hidden == 256 and rows % 64 == 0


def mlp_chain_bf16(x, w_splits, biases, auxs, outs, in_act, mid_act, aux_act) -> None:
    """csrc/pm_mlp.hip: up to four stacked 256 -> 256 layers of a ResidualMLP (two residual blocks, forward or data
    gradient) in one launch; biases / auxs may be None (see include/pmhip.h)"""
    L = len(w_splits)
    R, hid = x.shape[0], x.shape[1]

    def arr(ts):
        if ts is None:
            return None
        return (C.c_void_p * L)(*[(t.data_ptr() if t is not None else None) for t in ts])

    for t in list(outs) + [a for a in (auxs or []) if a is not None]:
        _ptr(t)                                                   # contiguity / dtype checks
    work = {"flops": 2.0 * L * R * hid * hid, "bytes": _nbytes(x, *(auxs or []), *outs), "detail": f"B{R} {L}x({hid}->{hid})"}
    _call("pm_mlp_chain_bf16", _ptr(x), L, arr(w_splits), arr(biases), arr(auxs), arr(outs), R, hid, in_act, mid_act, aux_act,
          LEAKY_SLOPE, tag="mlp_chain_bf16_kernel", work=work)


def layernorm_fwd(x, res, y, out, rstd, eps: float = 1e-5) -> None:
    """y = LayerNorm(x) over the last axis (no scale / offset); out = y + res when given (csrc/pm_mlp.hip)"""
    H = x.shape[-1]
    _call("pm_layernorm_fwd", _ptr(x), _ptr(res), _ptr(y), _ptr(out), _ptr(rstd), x.numel() // H, H, eps,
          tag="layernorm_fwd_kernel", work={"bytes": _nbytes(x, res, y, out)})


def layernorm_bwd(y, rstd, dy, dx) -> None:
    H = y.shape[-1]
    _call("pm_layernorm_bwd", _ptr(y), _ptr(rstd), _ptr(dy), _ptr(dx), y.numel() // H, H, tag="layernorm_bwd_kernel",
          work={"bytes": _nbytes(y, dy, dx)})


def relu_mask_fwd(x, mask, out) -> None:
    _call("pm_relu_mask_fwd", _ptr(x), _ptr(mask), _ptr(out), x.numel())


def relu_mask_bwd(x, mask, dout, dx) -> None:
    _call("pm_relu_mask_bwd", _ptr(x), _ptr(mask), _ptr(dout), _ptr(dx), x.numel())


# ---- PM-VAE evaluation paths (csrc/pm_eval.hip) ------------------------------------------------------------
def repeat_rows(src, dst, S: int) -> None:
    """dst[b*S + s, :] = src[b, :]"""
    B = src.shape[0]
    _call("pm_repeat_rows", _ptr(src), _ptr(dst), B, S, src.numel() // B)


def sigmoid(inp, out) -> None:
    _call("pm_sigmoid", _ptr(inp), _ptr(out), inp.numel())


def bernoulli_ll_rep_fwd(logits, x, w, ll, S: int) -> None:
    B = x.shape[0]
    D = x.numel() // B
    _call("pm_bernoulli_ll_rep_fwd", _ptr(logits), _ptr(x), _ptr(w), _ptr(ll), B, S, D, (w.numel() // B) if w is not None else D)


def normal_ll_rep_fwd(loc, x, log_scale, w, ll, S: int, scale_eps: float = 0.0) -> None:
    B = x.shape[0]
    D = x.numel() // B
    _call("pm_normal_ll_rep_fwd", _ptr(loc), _ptr(x), _ptr(log_scale), _ptr(w), _ptr(ll), B, S, D,
          (w.numel() // B) if w is not None else D, float(scale_eps))


def std_normal_logprob(z, lp) -> None:
    _call("pm_std_normal_logprob", _ptr(z), _ptr(lp), z.shape[0], z.shape[1])


def logmeanexp3(a, b, c, out, S: int, sample_major: bool = False) -> None:
    """out[b] = log mean_s exp(a + b - c); operands [B, S], or [S, B] with sample_major"""
    _call("pm_logmeanexp3", _ptr(a), _ptr(b), _ptr(c), _ptr(out), out.numel(), S, int(sample_major))


def gmm_sample_step(head, gumbel, eps, z, nc: int, i: int) -> None:
    R, k = z.shape
    _call("pm_gmm_sample_step", _ptr(head), _ptr(gumbel), _ptr(eps), _ptr(z), R, k, nc, i)


def diag_logprob_acc(params, z, out, P: int, sign: float = 1.0) -> None:
    """out[b] += sign * sum over the P positions of example b of log N(z; params[..., :Z], softplus(params[..., Z:2Z]) + 1e-5)"""
    Z = z.shape[-1]
    _call("pm_diag_logprob_acc", _ptr(params), params.shape[-1], _ptr(z), _ptr(out), z.numel() // Z, Z, P, float(sign))


def segment_wsum(v, w, out, sign: float = 1.0, accumulate: bool = False) -> None:
    B = out.numel()
    _call("pm_segment_wsum", _ptr(v), _ptr(w), _ptr(out), B, v.numel() // B, float(sign), int(accumulate))


def image_mask_mixture(mask, comps, seed: int, step_dev=None, stream_id: int = 0, desc_out=None, pattern_state=None,
                       pattern_refresh: int = 0) -> None:
    """mask [B,H,W,1] f32; comps: ctypes array of _lib.MaskComponent (host memory, copied into the launch);
    pattern_state: int64 [2] device tensor (noise epoch, pixels handed out) when a PATTERN component is present"""
    B, H, W = mask.shape[0], mask.shape[1], mask.shape[2]
    dptr = None
    if desc_out is not None:
        assert desc_out.is_cuda and desc_out.dtype == torch.int32 and desc_out.is_contiguous() and desc_out.numel() == 6 * B
        dptr = desc_out.data_ptr()
    sptr = None
    if pattern_state is not None:
        assert pattern_state.is_cuda and pattern_state.dtype == torch.int64 and pattern_state.numel() == 2
        sptr = pattern_state.data_ptr()
    _call("pm_image_mask_mixture", _ptr(mask), B, H, W, comps, len(comps), seed & (2 ** 64 - 1), _iptr(step_dev),
          stream_id, dptr, sptr, int(pattern_refresh))


def bernoulli_mask(mask, p: float, seed: int, step_dev=None, stream_id: int = 0) -> None:
    _call("pm_bernoulli_mask", _ptr(mask), mask.numel(), float(p), seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


def uniform_mask(mask, lo: int, span: int, seed: int, step_dev=None, stream_id: int = 0) -> None:
    B, D = mask.shape[0], mask.numel() // mask.shape[0]
    _call("pm_uniform_mask", _ptr(mask), B, D, int(lo), int(span), seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


def random_indices(idx, N: int, seed: int, step_dev=None, stream_id: int = 0) -> None:
    """idx int32 [B] <- uniform row indices in [0, N) (device Philox; a fresh draw per step)"""
    _call("pm_random_indices", _iptr(idx), idx.numel(), int(N), seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


def gather_u8_rows(src_u8, idx, dst, scale: float = 1.0) -> None:
    """dst [B, ...] f32 <- scale * src_u8[idx] (uint8 dataset resident in HBM; idx None: the first B rows)"""
    assert src_u8.is_cuda and src_u8.dtype == torch.uint8 and src_u8.is_contiguous()
    B = dst.shape[0]
    D = dst.numel() // B
    assert src_u8.numel() // src_u8.shape[0] == D
    _call("pm_gather_u8_rows", src_u8.data_ptr(), _iptr(idx), _ptr(dst), B, D, float(scale),
          work={"bytes": float(B * D * 5)})


def gumbel_fill(out, seed: int, step_dev=None, stream_id: int = 0) -> None:
    _call("pm_gumbel_fill", _ptr(out), out.numel(), seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


def dropout_mask(out, rate: float, seed: int, step_dev, stream_id: int = 0) -> None:
    _call("pm_dropout_mask", _ptr(out), out.numel(), rate, seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


# ---- VDVAE row-wise pieces (reference vdvae.py) ---------------------------------------------------------
def gelu_fwd(a, b, out) -> None:
    Ca, Cb = a.shape[-1], (b.shape[-1] if b is not None else 0)
    _call("pm_gelu_fwd", _ptr(a), _ptr(b), _ptr(out), a.numel() // Ca, Ca, Cb, work={"bytes": _nbytes(a, b, out)})


def gelu_bwd(a, b, dout, da, db, accumulate: bool) -> None:
    Ca, Cb = a.shape[-1], (b.shape[-1] if b is not None else 0)
    _call("pm_gelu_bwd", _ptr(a), _ptr(b), _ptr(dout), _ptr(da), _ptr(db), a.numel() // Ca, Ca, Cb, int(accumulate),
          work={"bytes": _nbytes(a, b, dout, da, db)})


def avgpool_fwd(x, out, k: int) -> None:
    B, H, W, C_ = x.shape
    _call("pm_avgpool_fwd", _ptr(x), _ptr(out), B, H, W, C_, k)


def avgpool_bwd(dout, dx, k: int) -> None:
    B, H, W, C_ = dx.shape
    _call("pm_avgpool_bwd", _ptr(dout), _ptr(dx), B, H, W, C_, k)


def resize_nearest_add(src, dst) -> None:
    """dst [B,H,W,C] += nearest-resized src [B,h,w,Cs][..., :C]"""
    B, h, w, Cs = src.shape
    _, H, W, C_ = dst.shape
    _call("pm_resize_nearest_add", _ptr(src), _ptr(dst), B, h, w, Cs, H, W, C_)


def resize_nearest_add_bwd(ddst, dsrc) -> None:
    B, h, w, Cs = dsrc.shape
    _, H, W, C_ = ddst.shape
    _call("pm_resize_nearest_add_bwd", _ptr(ddst), _ptr(dsrc), B, h, w, Cs, H, W, C_)


def broadcast_rows(src, dst) -> None:
    _call("pm_broadcast_rows", _ptr(src), _ptr(dst), dst.numel() // src.numel(), src.numel())


def add_cols(a, b, bcol: int, out) -> None:
    C_ = a.shape[-1]
    _call("pm_add_cols", _ptr(a), _ptr(b), b.shape[-1], bcol, _ptr(out), a.numel() // C_, C_)


def copy_cols(src, dst, dcol: int) -> None:
    C_ = src.shape[-1]
    _call("pm_copy_cols", _ptr(src), _ptr(dst), dst.shape[-1], dcol, src.numel() // C_, C_)


def scale_shift(x, a: float, c: float, out) -> None:
    _call("pm_scale_shift", _ptr(x), a, c, _ptr(out), x.numel())


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def vdvae_block_fused_ok(B: int, H: int, W: int, cin: int, cout: int, mid: int, k3: int) -> bool:
    """preconditions of pm_vdvae_block_fwd / _bwd (include/pmhip.h)"""
    return cin % 8 == 0 and cout % 8 == 0 and mid % 8 == 0 and mid <= 48 and W <= 62 and k3 in (1, 3)


def _dp(t):
    return t.data_ptr() if t is not None else None


def vdvae_block_io(x, hs, gs, out, wsplits, biases=None, x2=None, res=None, xpre=None, xg_out=None, backward: bool = False,
                   dense_k3: bool = False):
    """one Block's operands for vdvae_blocks_fwd / _bwd (struct pm_vdvae_block_io).  fwd: x = gelu(input) (or the raw input
    with xg_out), gs = g1..g3, out = Block output;  bwd: x = dout, gs = dh1..dh3, out = dxg."""
    io = _lib.VdvaeBlockIO()
    io._keep = (x, x2, res, xpre, xg_out, hs, gs, out, wsplits, biases)      # the tensors outlive the launch plan entry
    if backward:
        io.Cin, io.Cout, io.Ca = out.shape[-1], x.shape[-1], x.shape[-1]
    else:
        io.Ca = x.shape[-1]
        io.Cin, io.Cout = io.Ca + (x2.shape[-1] if x2 is not None else 0), out.shape[-1]
    io.x, io.x2, io.res, io.xpre, io.xg_out = _ptr(x), _ptr(x2), _ptr(res), _ptr(xpre), _ptr(xg_out)
    for i in range(3):
        io.h[i], io.g[i] = _ptr(hs[i]), _ptr(gs[i])
    io.out = _ptr(out)
    io.dense_k3 = int(dense_k3)
    for i in range(4):
        io.w[i] = wsplits[i].data_ptr()
        io.plane[i] = wsplits[i].numel() // 2
        io.bias[i] = _ptr(biases[i]) if biases is not None else None
    return io


def _blocks_call(fname, ios, B, H, W, mid, k3, tag, flops, nbytes):
    arr = (_lib.VdvaeBlockIO * len(ios))(*ios)
    _call(fname, arr, len(ios), B, H, W, mid, k3, tag=tag,
          work={"flops": flops, "bytes": nbytes, "detail": f"B{B} {H}x{W} mid{mid} k{k3} x{len(ios)}"})


def vdvae_blocks_fwd(ios, B: int, H: int, W: int, mid: int, k3: int) -> None:
    """csrc/pm_vdvae_block.hip: c1..c4 of up to four independent Blocks of one geometry in ONE launch"""
    flops = sum(2.0 * B * H * W * (io.Cin * mid + 2 * k3 * k3 * mid * mid + mid * io.Cout) for io in ios)
    nbytes = sum(4.0 * B * H * W * (io.Cin + io.Cout + 6 * mid) for io in ios)
    _blocks_call("pm_vdvae_blocks_fwd", ios, B, H, W, mid, k3, "vdvae_block_fwd_kernel", flops, nbytes)


def vdvae_blocks_bwd(ios, B: int, H: int, W: int, mid: int, k3: int) -> None:
    """the four data gradients of up to four independent Blocks in ONE launch"""
    flops = sum(2.0 * B * H * W * (io.Cin * mid + 2 * k3 * k3 * mid * mid + mid * io.Cout) for io in ios)
    nbytes = sum(4.0 * B * H * W * (io.Cin + io.Cout + 6 * mid) for io in ios)
    _blocks_call("pm_vdvae_blocks_bwd", ios, B, H, W, mid, k3, "vdvae_block_bwd_kernel", flops, nbytes)


def diag_sample_kl_fwd(post, prior, eps, z, kl, P: int) -> None:
    Z = eps.shape[-1]
    _call("pm_diag_sample_kl_fwd", _ptr(post), _ptr(prior), prior.shape[-1], _ptr(eps), _ptr(z), _ptr(kl), eps.numel() // Z, Z, P)


def diag_sample_kl_bwd(post, prior, eps, dz, g_kl: float, dpost, dprior) -> None:
    Z = eps.shape[-1]
    _call("pm_diag_sample_kl_bwd", _ptr(post), _ptr(prior), prior.shape[-1], _ptr(eps), _ptr(dz), g_kl, _ptr(dpost),
          _ptr(dprior), eps.numel() // Z, Z)


def sample_project_fwd(post, prior, eps, x_in, wz, bz, z, x2, kl, P: int) -> None:
    """x += h; z sampled + kl; x += z_proj(z) in one launch (csrc/pm_vdvae.hip, reference vdvae.py:558-562)"""
    Z, W = eps.shape[-1], x_in.shape[-1]
    _call("pm_sample_project_fwd", _ptr(post), _ptr(prior), prior.shape[-1], _ptr(eps), _ptr(x_in), _ptr(wz), _ptr(bz), _ptr(z),
          _ptr(x2), _ptr(kl), eps.numel() // Z, Z, W, P)


def sample_project_bwd(post, prior, eps, dx2, wz, g_kl: float, dpost, dprior) -> None:
    Z, W = eps.shape[-1], dx2.shape[-1]
    _call("pm_sample_project_bwd", _ptr(post), _ptr(prior), prior.shape[-1], _ptr(eps), _ptr(dx2), _ptr(wz), g_kl, _ptr(dpost),
          _ptr(dprior), eps.numel() // Z, Z, W)


def diag_tril_kl_fwd(post, mp, kl, Z: int, P: int) -> None:
    _call("pm_diag_tril_kl_fwd", _ptr(post), _ptr(mp), _ptr(kl), post.numel() // (2 * Z), Z, P)


def diag_tril_kl_bwd(post, mp, g: float, dmp, Z: int, P: int) -> None:
    _call("pm_diag_tril_kl_bwd", _ptr(post), _ptr(mp), g, _ptr(dmp), post.numel() // (2 * Z), Z, P)


def affine_fwd(x, gain, bias, out) -> None:
    C_ = x.shape[-1]
    _call("pm_affine_fwd", _ptr(x), _ptr(gain), _ptr(bias), _ptr(out), x.numel() // C_, C_)


def affine_bwd(x, gain, dout, dx, dgain, dbias) -> None:
    """dgain / dbias inside a ParamStore's gradient buffer: per-workgroup partial sums (no atomics; partials.PartialSums)"""
    from . import partials

    C_ = x.shape[-1]
    rows = x.numel() // C_
    own = partials.owner_of(dgain)
    ns = C.c_int()
    if (own is not None and partials.owner_of(dbias) is own and dgain.numel() == C_ and dbias.numel() == C_
            and _lib.load().pm_affine_bwd_part_slots(rows, C.byref(ns)) == 0):
        gbuf, goff = own.arena(own.offset(dgain), C_, ns.value)
        bbuf, boff = own.arena(own.offset(dbias), C_, ns.value)
        stride = own.entries[(own.offset(dgain), C_)][2]
        assert own.entries[(own.offset(dbias), C_)][2] == stride
        _call("pm_affine_bwd_part", _ptr(x), _ptr(gain), _ptr(dout), _ptr(dx), gbuf.data_ptr() + 4 * goff,
              bbuf.data_ptr() + 4 * boff, stride, ns.value, rows, C_)
        return
    _call("pm_affine_bwd", _ptr(x), _ptr(gain), _ptr(dout), _ptr(dx), _ptr(dgain), _ptr(dbias), rows, C_)


def dmol_ll_fwd(params, value, ll, nm: int, P: int, low: float = 0.0, high: float = 255.0) -> None:
    _call("pm_dmol_ll_fwd", _ptr(params), _ptr(value), _ptr(ll), value.numel(), nm, P, low, high)


def dmol_ll_bwd(params, value, g: float, dparams, nm: int, P: int, low: float = 0.0, high: float = 255.0) -> None:
    _call("pm_dmol_ll_bwd", _ptr(params), _ptr(value), g, _ptr(dparams), value.numel(), nm, P, low, high)


def dmol_mean(params, out, nm: int, low: float = 0.0, high: float = 255.0) -> None:
    _call("pm_dmol_mean", _ptr(params), _ptr(out), out.numel(), nm, low, high)


def vdvae_loss(rec, kl, pm_kl, num_dims: float, out) -> None:
    _call("pm_vdvae_loss", _ptr(rec), _ptr(kl), _ptr(pm_kl), rec.numel(), num_dims, _ptr(out))


def sumsq(x, out) -> None:
    _call("pm_sumsq", _ptr(x), x.numel(), _ptr(out))


def sumsq_det(x, out, scratch) -> None:
    """sum x^2 added in a fixed order (scratch: >= 1026 floats, zero before its first use)"""
    assert scratch.numel() >= 1026
    _call("pm_sumsq_det", _ptr(x), x.numel(), _ptr(out), _ptr(scratch))


def adam_step_clip_ema(p, g, m, v, ema, n_decay, count_dev, gnorm_sq, cfg: _lib.AdamCfg, clip: float, ema_rate: float,
                       skip_nonfinite: bool) -> None:
    _call("pm_adam_step_clip_ema", _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(), n_decay, _iptr(count_dev),
          _ptr(gnorm_sq), C.byref(cfg), clip, ema_rate, int(skip_nonfinite))


def pmvae_loss(rec, kl, mll, cfg: _lib.LossCfg, step_dev, out, g_rec, g_kl, g_mll) -> None:
    _call("pm_pmvae_loss", _ptr(rec), _ptr(kl), _ptr(mll), rec.shape[0], C.byref(cfg), _iptr(step_dev), _ptr(out), _ptr(g_rec), _ptr(g_kl), _ptr(g_mll))


# ---- VaDE mixture prior (reference vade.py) -----------------------------------------------------------------------------
def vade_prior_fwd(z, mu, log_scale, logits, lp) -> None:
    C_, k = mu.shape
    _call("pm_vade_prior_fwd", _ptr(z), _ptr(mu), _ptr(log_scale), _ptr(logits), _ptr(lp), z.numel() // k, k, C_)


def vade_prior_bwd(z, mu, log_scale, logits, g, dz, dmu, dlog_scale, dlogits) -> None:
    C_, k = mu.shape
    _call("pm_vade_prior_bwd", _ptr(z), _ptr(mu), _ptr(log_scale), _ptr(logits), _ptr(g), _ptr(dz), _ptr(dmu), _ptr(dlog_scale),
          _ptr(dlogits), z.numel() // k, k, C_)


def vade_cluster_probs(z, mu, log_scale, logits, probs, S: int) -> None:
    C_, k = mu.shape
    _call("pm_vade_cluster_probs", _ptr(z), _ptr(mu), _ptr(log_scale), _ptr(logits), _ptr(probs), z.numel() // k // S, S, k, C_)


def pmvae_loss_grads(B: int, cfg: _lib.LossCfg, step_dev, g_rec, g_kl, g_mll) -> None:
    """only the upstream gradients of pm_pmvae_loss (functions of the step counter, not of the forward pass)"""
    _call("pm_pmvae_loss", None, None, None, B, C.byref(cfg), _iptr(step_dev), None, _ptr(g_rec), _ptr(g_kl), _ptr(g_mll))


def adam_step(p, g, m, v, n_decay, count_dev, cfg: _lib.AdamCfg) -> None:
    passes = 8.0 if cfg.zero_grad else 7.0                      # reads p, g, m, v; writes p, m, v (and zeroes g)
    _call("pm_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), n_decay, _iptr(count_dev), C.byref(cfg),
          tag="adam_kernel<false, true>", work={"bytes": passes * 4.0 * p.numel()})


def adam_step_jobs(table, njobs: int, p, g, m, v, n_decay, count_dev, cfg: _lib.AdamCfg, part_bytes: float = 0.0) -> None:
    """pm_adam_step reading the weight-gradient kernels' partial sums itself (partials.PartialSums.adam_table)"""
    passes = 8.0 if cfg.zero_grad else 7.0
    _call("pm_adam_step_jobs", table.data_ptr(), njobs, _ptr(p), _ptr(g), _ptr(m), _ptr(v), n_decay, _iptr(count_dev),
          C.byref(cfg), tag="adam_jobs_kernel", work={"bytes": passes * 4.0 * p.numel() + part_bytes})


def counter_snapshot_increment(count_dev, snapshot_dev) -> None:
    _call("pm_counter_snapshot_increment", _iptr(count_dev), _iptr(snapshot_dev))


def counter_increment(count_dev) -> None:
    _call("pm_counter_increment", _iptr(count_dev))


def normal_fill(out, seed: int, step_dev, stream_id: int = 0) -> None:
    _call("pm_normal_fill", _ptr(out), out.numel(), seed & (2 ** 64 - 1), _iptr(step_dev), stream_id)


def split_weights(flat_params, out_bf16, jobs_dev, njobs: int, total_blocks: int) -> None:
    _call("pm_split_weights", _ptr(flat_params), out_bf16.data_ptr(), jobs_dev.data_ptr(), njobs, total_blocks,
          tag="split_weights_kernel", work={"bytes": float(out_bf16.numel()) * (2 + 2)})    # f32 read once per hi/lo pair, two bf16 writes


def fill_zero(t) -> None:
    _call("pm_fill_zero", t.data_ptr(), t.numel() * t.element_size())


_colsum_deferred: Optional[list] = None      # (x, arena address, stride, M, N, nslots) of column sums waiting for ONE launch


def colsum_defer_begin() -> None:
    """column sums into a ParamStore's gradient buffer issued from here on wait for colsum_defer_flush(): their inputs must
    stay untouched until then (a decoder's per-layer dpre buffers do)"""
    global _colsum_deferred
    if _colsum_deferred is None and not os.environ.get("PM_NO_COLSUM_MULTI"):
        _colsum_deferred = []


def colsum_defer_flush() -> None:
    """launches the deferred column sums: one pm_colsum_part_multi per 8 of them, on the current stream"""
    global _colsum_deferred
    jobs, _colsum_deferred = _colsum_deferred or [], None
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        if len(chunk) == 1:
            x, part, stride, M, N, ns = chunk[0]
            _call("pm_colsum_part", _ptr(x), M, N, part, stride, ns, tag="colsum_part_kernel",
                  work={"bytes": _nbytes(x) + 4.0 * ns * N})
            continue
        arr = (_lib.ColsumJob * len(chunk))()
        for j, (x, part, stride, M, N, ns) in enumerate(chunk):
            arr[j].x, arr[j].part, arr[j].part_stride, arr[j].M, arr[j].N, arr[j].nslots = x.data_ptr(), part, stride, M, N, ns
        _call("pm_colsum_part_multi", arr, len(chunk), tag="colsum_part_multi_kernel",
              work={"bytes": sum(_nbytes(x) + 4.0 * ns * N for x, _, _, _, N, ns in chunk)})


def colsum(x, out) -> None:
    """out[n] += sum_m x[m, n].  `out` inside a ParamStore's gradient buffer: per-workgroup partial sums (no atomics,
    deterministic; partials.PartialSums), added by the store's next reduce."""
    from . import partials

    N = x.shape[-1]
    M = x.numel() // N
    own = partials.owner_of(out)
    ns = C.c_int()
    vec = N % 4 == 0 and 1024 % N == 0
    if (own is not None and (x.data_ptr() % 16 == 0 or not vec) and out.numel() == N
            and _lib.load().pm_colsum_part_slots(M, N, C.byref(ns)) == 0):
        buf, off = own.arena(own.offset(out), N, ns.value)
        stride = own.entries[(own.offset(out), N)][2]
        if _colsum_deferred is not None:
            _colsum_deferred.append((x, buf.data_ptr() + 4 * off, stride, M, N, ns.value))
            return
        _call("pm_colsum_part", _ptr(x), M, N, buf.data_ptr() + 4 * off, stride, ns.value, tag="colsum_part_kernel",
              work={"bytes": _nbytes(x) + 4.0 * ns.value * N})
        return
    _call("pm_colsum", _ptr(x), _ptr(out), M, N, tag="colsum_kernel", work={"bytes": _nbytes(x, out)})


def axpy1(x, y) -> None:
    _call("pm_axpy1", _ptr(x), _ptr(y), x.numel())


class Graph:
    """HIP graph of a launch sequence captured on the current stream."""

    def __init__(self):
        self._exec = C.c_void_p()

    def __enter__(self):
        _lib.check(_lib.load().pm_graph_begin(_stream()), "pm_graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = _lib.load().pm_graph_end(_stream(), C.byref(self._exec))
        if et is None:
            _lib.check(rc, "pm_graph_end")
        return False

    def launch(self) -> None:
        _lib.check(_lib.load().pm_graph_launch(self._exec, _stream()), "pm_graph_launch")

    def __del__(self):
        try:
            if self._exec:
                _lib.load().pm_graph_destroy(self._exec)
        except Exception:
            pass


class Event:
    """hipEvent_t recorded on the stream the kernels are launched on."""

    def __init__(self):
        self._ev = C.c_void_p()
        _lib.check(_lib.load().pm_event_create(C.byref(self._ev)), "pm_event_create")

    def record(self) -> None:
        _lib.check(_lib.load().pm_event_record(self._ev, _stream()), "pm_event_record")

    def synchronize(self) -> None:
        _lib.check(_lib.load().pm_event_synchronize(self._ev), "pm_event_synchronize")

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        _lib.check(_lib.load().pm_event_elapsed_ms(self._ev, end._ev, C.byref(ms)), "pm_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            _lib.load().pm_event_destroy(self._ev)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# lookahead posteriors (reference models/lookahead.py)
# ------------------------------------------------------------------------------------------
def lookahead_inputs(imp, b, inds, out) -> None:
    """imp [B,Z,P,C], b [B,P] (or [B,H,W,1]), inds int32 [S] -> out [B*Z*S, P, C+1] = [imp * b_look | b_look]"""
    B, Z = imp.shape[0], imp.shape[1]
    C_ = imp.shape[-1]
    P = imp.numel() // (B * Z * C_)
    S = inds.numel()
    assert b.numel() == B * P and out.numel() == B * Z * S * P * (C_ + 1)
    _call("pm_lookahead_inputs", _ptr(imp), _ptr(b), _iptr(inds), _ptr(out), B, Z, S, P, C_)


def lookahead_ll_fwd(params, inds, zs, b, ll) -> None:
    """params [B,F,2k], inds int32 [S], zs [B,Z,S,k], b [B,F] -> ll [B]"""
    B, F, k2 = params.shape
    Z, S = zs.shape[1], zs.shape[2]
    assert b.numel() == B * F and zs.shape[0] == B and zs.shape[3] * 2 == k2 and inds.numel() == S
    _call("pm_lookahead_ll_fwd", _ptr(params), _iptr(inds), _ptr(zs), _ptr(b), _ptr(ll), B, F, Z, S, k2 // 2)


def lookahead_ll_bwd(params, inds, zs, b, g, dparams) -> None:
    B, F, k2 = params.shape
    Z, S = zs.shape[1], zs.shape[2]
    assert dparams.shape == params.shape and b.numel() == B * F
    _call("pm_lookahead_ll_bwd", _ptr(params), _iptr(inds), _ptr(zs), _ptr(b), _ptr(g), _ptr(dparams), B, F, Z, S, k2 // 2)


def lookahead_info_gains(params, cur_ent, b, gains) -> None:
    """params [F,2k], cur_ent [1], b [F] -> gains [F]"""
    F, k2 = params.shape
    _call("pm_lookahead_info_gains", _ptr(params), _ptr(cur_ent), _ptr(b), _ptr(gains), F, k2 // 2)


def acquisition_policy(gains, probs, action) -> None:
    """gains [F] -> probs [F] (softmax of where(gains == -inf, -1e10, gains)), action int32 [1] (argmax, first among ties)"""
    _call("pm_acquisition_policy", _ptr(gains), _ptr(probs), _iptr(action), gains.numel())


def reconstruction_rmse(imp, x, b, recon, rmse) -> None:
    """imp [S, *x.shape], x, b (x's shape, or one mask channel) -> recon = mean_s imp, rmse [1] over the unobserved entries"""
    S, D = imp.shape[0], x.numel()
    assert imp.numel() == S * D and recon.numel() == D
    _call("pm_reconstruction_rmse", _ptr(imp), _ptr(x), _ptr(b), _ptr(recon), _ptr(rmse), S, D, x.shape[-1], b.shape[-1])
