"""ctypes binding of libmlpg_hip.so (include/mlpg_hip.h) + thin tensor helpers.

PyTorch is used here for device memory, streams and nothing else: every
numerical result comes from the hand-written HIP kernels behind the C ABI.
There is NO CPU fallback: if the shared library or a GPU is missing, the calls
raise.
"""
import ctypes
import os
import sys
import threading

import numpy as np

F32, F64 = 0, 1
ABI_VERSION = 14               # mlpg_hip_abi_version() of the library this binding was written for
VAR_FRAME, VAR_GLOBAL, VAR_UNIT = 0, 1, 2
ALGO_AUTO, ALGO_GENERIC, ALGO_WAVE, ALGO_STRIP, ALGO_PIPE, ALGO_CONST, ALGO_CHUNK, ALGO_FIR = 0, 1, 2, 3, 4, 5, 6, 7

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNMNKWII_AMD_SO selects another build of the same library (kernel experiments); default: the in-tree build
SO_PATH = os.environ.get("NNMNKWII_AMD_SO") or os.path.join(_HERE, "csrc", "libmlpg_hip.so")

EXPORTS = (
    "mlpg_hip_abi_version",
    "mlpg_hip_launch_count",
    "mlpg_hip_last_error",
    "mlpg_hip_device_count",
    "mlpg_hip_shutdown",
    "mlpg_hip_forward",
    "mlpg_hip_forward_host",
    "mlpg_hip_fastdtw_host",
    "mlpg_hip_host_alloc",
    "mlpg_hip_host_free",
    "mlpg_hip_forward_streams",
    "mlpg_hip_backward",
    "mlpg_hip_delta_features",
    "mlpg_hip_modspec",
    "mlpg_hip_inv_modspec",
    "mlpg_hip_modspec_smoothing",
    "mlpg_hip_modspec_backward",
    "mlpg_hip_modspec_set_direct",
    "mlpg_hip_trim_lengths",
    "mlpg_hip_fastdtw",
    "mlpg_hip_forward_host_multi",
    "mlpg_hip_fastdtw_host_multi",
    "mlpg_hip_host_chunk_plan",
    "mlpg_hip_dtw_level_windows",
    "mlpg_hip_dtw_level_from_costs",
    "mlpg_hip_fastdtw_l2",
    "mlpg_hip_gather_path",
    "mlpg_hip_gmm_convert",
    "mlpg_hip_stream_copy",
    "mlpg_hip_unit_mse_step",
    "mlpg_hip_unit_mse_workspace_bytes",
    "mlpg_hip_unit_mse_workspace_bytes_t",
    "mlpg_hip_unit_mse_form",
    "mlpg_hip_host_copy",
    "mlpg_hip_backward_host",
)


class HipExtensionError(RuntimeError):
    """The HIP extension is missing, failed to load, or a call into it failed."""


_lib = None
_lock = threading.Lock()


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own libamdhip64 / libhsa-runtime64 and loads them by path;
    libmlpg_hip.so asks for "libamdhip64.so.7".  Loaded after torch, the library is given torch's copy (same SONAME) and all is
    well; loaded BEFORE torch -- a program that starts with the numpy-only entry points -- it would bring in the system's runtime,
    torch would then add its own, and with two runtimes in the process ``torch.cuda.is_available()`` turns False.  So, when torch is
    installed but not imported yet, its runtime libraries are loaded first (no ``import torch``: that costs seconds and the
    host-pointer entry points do not need it)."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: without it the system's runtime is used, as before
        pass


def lib():
    """Load libmlpg_hip.so (once). Raises HipExtensionError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(SO_PATH):
            raise HipExtensionError(
                "nnmnkwii_amd: %s is missing -- build it with `python nnmnkwii_amd/csrc/build.py` "
                "(or __graft_entry__.build()); there is no CPU fallback" % SO_PATH)
        _preload_torch_hip_runtime()
        try:
            L = ctypes.CDLL(SO_PATH)
        except OSError as e:  # pragma: no cover
            raise HipExtensionError("nnmnkwii_amd: cannot load %s: %s" % (SO_PATH, e))
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.mlpg_hip_abi_version.restype = ci
        L.mlpg_hip_abi_version.argtypes = []
        L.mlpg_hip_launch_count.restype = ctypes.c_longlong
        L.mlpg_hip_launch_count.argtypes = [ci]
        L.mlpg_hip_last_error.restype = ctypes.c_char_p
        L.mlpg_hip_last_error.argtypes = []
        L.mlpg_hip_device_count.restype = ci
        L.mlpg_hip_device_count.argtypes = []
        L.mlpg_hip_shutdown.restype = None
        L.mlpg_hip_shutdown.argtypes = []
        L.mlpg_hip_forward.restype = ci
        L.mlpg_hip_forward.argtypes = [ci, vp, ci, ci, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        L.mlpg_hip_forward_host.restype = ci
        L.mlpg_hip_forward_host.argtypes = [ci, ci, ci, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        L.mlpg_hip_backward_host.restype = ci
        L.mlpg_hip_backward_host.argtypes = [ci, ci, ci, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        L.mlpg_hip_fastdtw_host.restype = ci
        L.mlpg_hip_fastdtw_host.argtypes = [ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_double, ci, ctypes.c_double,
                                            vp, vp, vp, vp, vp, vp]
        L.mlpg_hip_forward_host_multi.restype = ci
        L.mlpg_hip_forward_host_multi.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        L.mlpg_hip_fastdtw_host_multi.restype = ci
        L.mlpg_hip_fastdtw_host_multi.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_double, ci,
                                                  ctypes.c_double, vp, vp, vp, vp, vp, vp]
        L.mlpg_hip_host_chunk_plan.restype = ctypes.c_longlong
        L.mlpg_hip_host_chunk_plan.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ci, ctypes.c_longlong, vp, vp, vp, vp]
        L.mlpg_hip_host_alloc.restype = vp
        L.mlpg_hip_host_alloc.argtypes = [ctypes.c_size_t]
        L.mlpg_hip_host_copy.restype = ci
        L.mlpg_hip_host_copy.argtypes = [vp, vp, ctypes.c_size_t]
        L.mlpg_hip_host_free.restype = None
        L.mlpg_hip_host_free.argtypes = [vp]
        L.mlpg_hip_forward_streams.restype = ci
        L.mlpg_hip_forward_streams.argtypes = [ci, vp, ci, ci, vp, vp, ci, ctypes.c_int64, vp, ci, ci, ci, vp, ci, vp, vp, vp,
                                               vp, ctypes.c_int64, vp]
        L.mlpg_hip_backward.restype = ci
        L.mlpg_hip_backward.argtypes = [ci, vp, ci, ci, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        L.mlpg_hip_delta_features.restype = ci
        L.mlpg_hip_delta_features.argtypes = [ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp]
        L.mlpg_hip_modspec.restype = ci
        L.mlpg_hip_modspec.argtypes = [ci, vp, vp, ci, ci, ci, ci, ci, vp, vp]
        L.mlpg_hip_inv_modspec.restype = ci
        L.mlpg_hip_inv_modspec.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, vp]
        L.mlpg_hip_modspec_smoothing.restype = ci
        L.mlpg_hip_modspec_smoothing.argtypes = [ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
        L.mlpg_hip_modspec_backward.restype = ci
        L.mlpg_hip_modspec_backward.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.mlpg_hip_modspec_set_direct.restype = None
        L.mlpg_hip_modspec_set_direct.argtypes = [ci]
        L.mlpg_hip_trim_lengths.restype = ci
        L.mlpg_hip_trim_lengths.argtypes = [ci, vp, ci, vp, ci, ci, ci, cd, vp]
        L.mlpg_hip_fastdtw_l2.restype = ci
        L.mlpg_hip_fastdtw_l2.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
        L.mlpg_hip_dtw_level_windows.restype = ci
        L.mlpg_hip_dtw_level_windows.argtypes = [ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci]
        L.mlpg_hip_dtw_level_from_costs.restype = ci
        L.mlpg_hip_dtw_level_from_costs.argtypes = [ci, vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, vp, ctypes.c_longlong, vp, vp, vp,
                                                    ci, vp]
        L.mlpg_hip_fastdtw.restype = ci
        L.mlpg_hip_fastdtw.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, ci, vp, vp, vp, vp]
        L.mlpg_hip_gmm_convert.restype = ci
        L.mlpg_hip_gmm_convert.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, vp]
        L.mlpg_hip_gather_path.restype = ci
        L.mlpg_hip_gather_path.argtypes = [ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.mlpg_hip_unit_mse_step.restype = ci
        L.mlpg_hip_unit_mse_step.argtypes = [ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, cd, vp, vp, vp, vp, vp,
                                             ctypes.c_size_t]
        L.mlpg_hip_unit_mse_workspace_bytes.restype = ctypes.c_size_t
        L.mlpg_hip_unit_mse_workspace_bytes.argtypes = [ci, ci, ci]
        L.mlpg_hip_unit_mse_workspace_bytes_t.restype = ctypes.c_size_t
        L.mlpg_hip_unit_mse_workspace_bytes_t.argtypes = [ci, ci, ci, ci]
        L.mlpg_hip_unit_mse_form.restype = ci
        L.mlpg_hip_unit_mse_form.argtypes = [ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
        L.mlpg_hip_stream_copy.restype = ci
        L.mlpg_hip_stream_copy.argtypes = [ci, vp, vp, vp, ctypes.c_size_t]
        if L.mlpg_hip_abi_version() != ABI_VERSION:
            raise HipExtensionError("nnmnkwii_amd: %s has ABI version %d, this binding needs %d -- rebuild it with "
                                    "`python nnmnkwii_amd/csrc/build.py`" % (SO_PATH, L.mlpg_hip_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HipExtensionError("%s failed (%d): %s" % (what, rc, lib().mlpg_hip_last_error().decode()))


def torch_mod():
    import torch
    return torch


_gpu_ok = False   # torch.cuda.is_available() has been seen True and the library is loaded (neither changes afterwards)


def require_gpu(device=None):
    """Return a torch.device for the GPU to use, or raise (no CPU fallback)."""
    global _gpu_ok
    if _gpu_ok and device is not None and device.__class__.__name__ == "device" and device.type == "cuda" and device.index is not None:
        return device            # the hot path: a tensor's own device (torch.cuda.is_available() alone costs microseconds)
    torch = torch_mod()
    lib()
    if not torch.cuda.is_available():
        raise HipExtensionError("nnmnkwii_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                                "there is no CPU fallback")
    _gpu_ok = True
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise HipExtensionError("nnmnkwii_amd computes on GPU only, got device %s" % device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class PackedWindows(tuple):
    """(int32 l[], int32 u[], float64 coeff[], num_windows): what pack_windows returns; pass it instead of the window
    list to skip the packing in a hot loop.  `ptrs()`: the three host addresses as ints (computed once: numpy's
    `.ctypes.data_as` costs microseconds per call, the C ABI takes the address itself)."""

    def ptrs(self):
        pt = self.__dict__.get("_ptrs")
        if pt is None:
            pt = self.__dict__["_ptrs"] = (self[0].ctypes.data, self[1].ctypes.data, self[2].ctypes.data)
        return pt


def _win_args(windows):
    """(num_windows, address of l[], u[], coeff[], keep-alive) for the C ABI."""
    if isinstance(windows, PackedWindows):
        a, b, c = windows.ptrs()
        return windows[3], a, b, c, windows
    wl, wu, wc = pack_windows(windows)
    return len(wl), wl.ctypes.data, wu.ctypes.data, wc.ctypes.data, (wl, wu, wc)


def pack_windows(windows):
    """Reference `(l, u, coeff)` triples -> (int32 l[], int32 u[], float64 coeff[]) host arrays."""
    if isinstance(windows, PackedWindows):
        return windows[0], windows[1], windows[2]
    wl, wu, wc = [], [], []
    for l, u, coeff in windows:
        l, u = int(l), int(u)
        assert l >= 0 and u >= 0                     # paramgen/_mlpg.py:44
        coeff = np.asarray(coeff, dtype=np.float64).ravel()
        assert len(coeff) == l + u + 1               # paramgen/_mlpg.py:45
        wl.append(l)
        wu.append(u)
        wc.append(coeff)
    return (np.ascontiguousarray(wl, dtype=np.int32), np.ascontiguousarray(wu, dtype=np.int32),
            np.ascontiguousarray(np.concatenate(wc)))


def prepack_windows(windows):
    """Pack once, reuse: the result is accepted wherever a window list is."""
    if isinstance(windows, PackedWindows):
        return windows
    wl, wu, wc = pack_windows(windows)
    return PackedWindows((wl, wu, wc, len(wl)))


def _nw(windows):
    return windows[3] if isinstance(windows, PackedWindows) else len(windows)


_WIN_CACHE = {}     # id(window list) -> (the list, its signature, PackedWindows): the literal per-utterance calls pass the same list


def cached_windows(windows):
    """PackedWindows of a window list, remembered per list OBJECT (a loop of paramgen.mlpg calls passes the same list every
    time: packing costs 5 us, a config-1 call 30).  The remembered tables are used only if the list still holds the same
    (l, u, coefficient object) triples with the same coefficient bytes; anything else is packed afresh."""
    if isinstance(windows, PackedWindows):
        return windows
    ent = _WIN_CACHE.get(id(windows))
    try:
        if ent is not None and ent[0] is windows and len(windows) == len(ent[1]):
            for w, g in zip(windows, ent[1]):
                c = w[2]
                if w[0] != g[0] or w[1] != g[1] or c is not g[2] or (c.tobytes() if type(c) is np.ndarray else tuple(c)) != g[3]:
                    break
            else:
                return ent[2]
        packed = prepack_windows(windows)
        sig = tuple((w[0], w[1], w[2], w[2].tobytes() if type(w[2]) is np.ndarray else tuple(w[2])) for w in windows)
    except (TypeError, IndexError):      # not a list of (l, u, coeff) triples this cache understands: no caching
        return prepack_windows(windows)
    if len(_WIN_CACHE) >= 64:
        _WIN_CACHE.clear()
    _WIN_CACHE[id(windows)] = (windows, sig, packed)
    return packed


def _dt(t):
    torch = torch_mod()
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise HipExtensionError("unsupported dtype %s (float32/float64 only)" % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()      # (argtypes are declared: ctypes converts the int itself)


def _np(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_raw_stream = None


def _stream(device):
    """The current torch stream of `device` as the raw hipStream_t (an int).  torch._C._cuda_getCurrentRawStream is the
    cheap way (what torch's own compiled code uses); torch.cuda.current_stream(...) builds a Stream object per call."""
    global _raw_stream
    if _raw_stream is None:
        torch = torch_mod()
        f = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        _raw_stream = f if f is not None else (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    return _raw_stream(device.index if device.index is not None else torch_mod().cuda.current_device())


def forward(mean, var, windows, lengths=None, algo=ALGO_AUTO, want_status=True):
    """Batched MLPG on device tensors.

    mean (B, T, D) cuda float32/float64 contiguous; var: same shape, (D,) or None
    (unit variances); lengths: cuda int32 (B,) or None.  Returns (out (B,T,sd),
    status int32 (B*sd) or None), both on the device, enqueued on the current
    stream (no synchronisation).
    """
    torch = torch_mod()
    assert mean.is_cuda and mean.dim() == 3 and mean.is_contiguous()
    B, T, D = mean.shape
    nw, pl, pu, pc, _keep = _win_args(windows)
    if var is None:
        mode = VAR_UNIT
    elif var.dim() == 1:
        mode = VAR_GLOBAL
        assert var.shape[0] == D and var.dtype == mean.dtype and var.is_contiguous() and var.device == mean.device
    else:
        mode = VAR_FRAME
        assert var.shape == mean.shape and var.dtype == mean.dtype and var.is_contiguous()
        assert var.device == mean.device
    if lengths is not None:
        assert lengths.dtype == torch.int32 and lengths.shape == (B,) and lengths.device == mean.device
    out = torch.empty((B, T, D // nw), dtype=mean.dtype, device=mean.device)
    status = torch.empty((B * (D // nw),), dtype=torch.int32, device=mean.device) if want_status else None
    rc = lib().mlpg_hip_forward(mean.device.index, _stream(mean.device), _dt(mean), algo, _p(mean), _p(var), mode,
                                _p(lengths), B, T, D, nw, pl, pu, pc, _p(out), _p(status))
    _check(rc, "mlpg_hip_forward")
    return out, status


def current_device_index(device=None):
    """GPU index for the host-pointer entry points: an explicit int / "cuda:1" / object with .index, otherwise the
    process's current device (what torch.cuda.set_device(local_rank) selected in a one-process-per-GPU job; 0 when
    torch has not been imported or has not touched the GPU yet -- the entry points themselves need no torch)."""
    if device is not None:
        idx = device if isinstance(device, str) else getattr(device, "index", device)
        if isinstance(idx, str):
            idx = idx.split(":")[-1] if ":" in idx else None
        if idx is not None:
            return int(idx)
    t = sys.modules.get("torch")
    # (another thread may be in the middle of `import torch`: sys.modules then holds a module without its attributes yet)
    cuda = getattr(t, "cuda", None)
    ready = getattr(cuda, "is_initialized", None)
    if ready is not None and ready():
        return int(cuda.current_device())
    return 0


def device_list(device=None):
    """The int32 device list of the *_host_multi entry points: ``"all"`` -> empty (= every visible device), a list /
    tuple of indices as given (an index may repeat: each occurrence gets its own streams and staging buffers), anything
    else -> the one device current_device_index picks."""
    if isinstance(device, str) and device == "all":
        return np.zeros((0,), dtype=np.int32)
    if isinstance(device, (list, tuple, np.ndarray)):
        return np.ascontiguousarray([current_device_index(d) for d in device], dtype=np.int32)
    return np.asarray([current_device_index(device)], dtype=np.int32)


def host_chunk_plan(n_items, target_items, num_devices):
    """mlpg_hip_host_chunk_plan (pure host logic, no GPU needed): how the host-memory calls deal a batch to the devices.
    Returns int arrays (entry, slot, first, count), one element per chunk."""
    L = lib()
    n = int(L.mlpg_hip_host_chunk_plan(int(n_items), int(target_items), int(num_devices), 0, None, None, None, None))
    if n < 0:
        _check(n, "mlpg_hip_host_chunk_plan")
    entry, slot = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    first, count = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    L.mlpg_hip_host_chunk_plan(int(n_items), int(target_items), int(num_devices), n, _np(entry), _np(slot), _np(first), _np(count))
    return entry, slot, first, count


_host_gpu_seen = False     # mlpg_hip_device_count() > 0 has been seen (it does not change afterwards)


def forward_host(mean, var, windows, lengths=None, algo=ALGO_AUTO, device=None):
    """Batched MLPG, numpy in -> numpy out through mlpg_hip_forward_host[_multi] (no torch involved): mean (B, T, D)
    float32/float64 C-contiguous, var same shape / (D,) / None, lengths int32 (B,) or None; device: GPU index
    (default: the current device, see current_device_index), a list of indices, or "all" (every visible device; the
    utterance chunks are dealt round-robin, see device_list).  One device: mlpg_hip_forward_host, whose small calls
    (a single utterance: the literal paramgen.mlpg call) take the library's short path.
    Returns (out (B, T, sd) ndarray, status int32 (B, sd))."""
    global _host_gpu_seen
    L = lib()
    if not _host_gpu_seen:
        if L.mlpg_hip_device_count() <= 0:
            raise HipExtensionError("nnmnkwii_amd needs an AMD GPU (none visible to the HIP runtime); there is no CPU fallback")
        _host_gpu_seen = True
    assert mean.ndim == 3 and mean.flags.c_contiguous and mean.dtype in (np.float32, np.float64)
    B, T, D = mean.shape
    pw = cached_windows(windows)
    nw = pw[3]
    pl, pu, pc = pw.ptrs()
    dt = F32 if mean.dtype == np.float32 else F64
    if var is None:
        mode, pv = VAR_UNIT, None
    else:
        assert var.dtype == mean.dtype and var.flags.c_contiguous
        mode = VAR_GLOBAL if var.ndim == 1 else VAR_FRAME
        assert var.shape == ((D,) if var.ndim == 1 else mean.shape)
        pv = var.ctypes.data
    plen = None
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        assert lengths.shape == (B,)
        plen = lengths.ctypes.data
    out = np.empty((B, T, D // nw), dtype=mean.dtype)
    status = np.zeros((B, D // nw), dtype=np.int32)
    if isinstance(device, (list, tuple, np.ndarray)) or (isinstance(device, str) and device == "all"):
        devs = device_list(device)
        rc = L.mlpg_hip_forward_host_multi(_np(devs) if len(devs) else None, len(devs), dt, algo, mean.ctypes.data, pv, mode, plen,
                                           B, T, D, nw, pl, pu, pc, out.ctypes.data, status.ctypes.data)
        _check(rc, "mlpg_hip_forward_host_multi")
    else:
        rc = L.mlpg_hip_forward_host(current_device_index(device), dt, algo, mean.ctypes.data, pv, mode, plen,
                                     B, T, D, nw, pl, pu, pc, out.ctypes.data, status.ctypes.data)
        if rc != 0:
            _check(rc, "mlpg_hip_forward_host")
    return out, status


def backward_host(var, grad_out, windows, D, out_dtype=np.float32, lengths=None, algo=ALGO_AUTO, device=None):
    """MLPG backward, numpy in -> numpy out through mlpg_hip_backward_host (no torch involved): grad_out (B, T, sd)
    float32/float64 C-contiguous, var (B, T, D) / (D,) of the same dtype / None (unit variances), lengths int32 (B,) or None.
    Returns (grad (B, T, D) of out_dtype, status int32 (B, sd)).  The literal paramgen.mlpg_grad call and the backward of
    autograd.MLPG on CPU tensors: the library's short path (see forward_host), in pieces of whole utterances."""
    global _host_gpu_seen
    L = lib()
    if not _host_gpu_seen:
        if L.mlpg_hip_device_count() <= 0:
            raise HipExtensionError("nnmnkwii_amd needs an AMD GPU (none visible to the HIP runtime); there is no CPU fallback")
        _host_gpu_seen = True
    assert grad_out.ndim == 3 and grad_out.flags.c_contiguous and grad_out.dtype in (np.float32, np.float64)
    B, T, sd = grad_out.shape
    pw = cached_windows(windows)
    nw = pw[3]
    assert D == nw * sd
    pl, pu, pc = pw.ptrs()
    dt = F32 if grad_out.dtype == np.float32 else F64
    out_dtype = np.dtype(out_dtype)
    assert out_dtype in (np.float32, np.float64)
    if var is None:
        mode, pv = VAR_UNIT, None
    else:
        assert var.dtype == grad_out.dtype and var.flags.c_contiguous
        mode = VAR_GLOBAL if var.ndim == 1 else VAR_FRAME
        assert var.shape == ((D,) if var.ndim == 1 else (B, T, D))
        pv = var.ctypes.data
    plen = None
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        assert lengths.shape == (B,)
        plen = lengths.ctypes.data
    grad = np.empty((B, T, D), dtype=out_dtype)
    status = np.zeros((B, sd), dtype=np.int32)
    rc = L.mlpg_hip_backward_host(current_device_index(device), dt, F32 if out_dtype == np.float32 else F64, algo, pv, mode,
                                  grad_out.ctypes.data, plen, B, T, D, nw, pl, pu, pc, grad.ctypes.data, status.ctypes.data)
    if rc != 0:
        _check(rc, "mlpg_hip_backward_host")
    return grad, status


class _PinnedOwner(object):
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                lib().mlpg_hip_host_free(ctypes.c_void_p(self.ptr))
        except Exception:  # interpreter shutdown
            pass


def pinned_empty(shape, dtype=np.float64):
    """An uninitialised numpy array in pinned (page-locked) host memory: forward_host transfers such arrays in place
    (no staging copy), at the PCIe rate."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    ptr = lib().mlpg_hip_host_alloc(max(n, 1))
    if not ptr:
        raise HipExtensionError("mlpg_hip_host_alloc failed: %s" % lib().mlpg_hip_last_error().decode())
    buf = (ctypes.c_char * max(n, 1)).from_address(ptr)
    buf._owner = _PinnedOwner(ptr)     # freed when the last array viewing this buffer goes away
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class StreamDesc(ctypes.Structure):
    """mlpg_hip_stream_t (include/mlpg_hip.h)."""
    _fields_ = [("in_col", ctypes.c_int32), ("out_col", ctypes.c_int32), ("static_dim", ctypes.c_int32),
                ("num_windows", ctypes.c_int32), ("win_first", ctypes.c_int32)]


def forward_streams(mean, var, streams, lengths=None, algo=ALGO_AUTO, want_status=True):
    """Multi-stream MLPG over one (B, T, ld) CUDA batch, streams consumed in place.

    ``streams``: list of ``(in_col, static_dim, windows)``; ``windows`` is a list of (l, u, coeff)
    triples, or None / [] for a pass-through stream (static columns copied).  Trajectories are
    written side by side in table order.  Returns (out (B, T, sum static_dim), status (B, sum
    static_dim) int32 or None).
    """
    torch = torch_mod()
    assert mean.is_cuda and mean.dim() == 3 and mean.is_contiguous()
    B, T, ld = mean.shape
    if var is None:
        mode = VAR_UNIT
    elif var.dim() == 1:
        mode = VAR_GLOBAL
        assert var.shape[0] == ld and var.dtype == mean.dtype and var.is_contiguous() and var.device == mean.device
    else:
        mode = VAR_FRAME
        assert var.shape == mean.shape and var.dtype == mean.dtype and var.is_contiguous() and var.device == mean.device
    if lengths is not None:
        assert lengths.dtype == torch.int32 and lengths.shape == (B,) and lengths.device == mean.device
    table = (StreamDesc * max(len(streams), 1))()
    wl_all, wu_all, wc_all = [], [], []
    seen = {}                       # window lists shared between streams are packed once
    out_col = 0
    for k, (in_col, sd, windows) in enumerate(streams):
        nw = len(windows) if windows else 0
        first = 0
        if nw:
            key = id(windows)
            if key not in seen:
                wl, wu, wc = pack_windows(windows)
                seen[key] = len(wl_all)
                wl_all.extend(wl.tolist())
                wu_all.extend(wu.tolist())
                wc_all.extend(wc.tolist())
            first = seen[key]
        table[k] = StreamDesc(int(in_col), out_col, int(sd), nw, first)
        out_col += int(sd)
    wl = np.ascontiguousarray(wl_all, dtype=np.int32)
    wu = np.ascontiguousarray(wu_all, dtype=np.int32)
    wc = np.ascontiguousarray(wc_all if wc_all else [0.0], dtype=np.float64)
    out = torch.empty((B, T, out_col), dtype=mean.dtype, device=mean.device)
    status = torch.empty((B, out_col), dtype=torch.int32, device=mean.device) if want_status else None
    rc = lib().mlpg_hip_forward_streams(mean.device.index, _stream(mean.device), _dt(mean), algo, _p(mean), _p(var), mode,
                                        ld, _p(lengths), B, T, len(streams), ctypes.addressof(table), len(wl_all),
                                        _np(wl), _np(wu), _np(wc), _p(out), out_col, _p(status))
    _check(rc, "mlpg_hip_forward_streams")
    return out, status


def backward(var, grad_out, windows, D, lengths=None, out_dtype=None, algo=ALGO_AUTO, want_status=True):
    """Batched MLPG gradient w.r.t. means on device tensors.

    grad_out (B, T, sd); var (B, T, D), (D,) or None (unit).  Returns
    (grad_mean (B, T, D) in out_dtype (default float32), status).
    """
    torch = torch_mod()
    assert grad_out.is_cuda and grad_out.dim() == 3 and grad_out.is_contiguous()
    B, T, sd = grad_out.shape
    nw, pl, pu, pc, _keep = _win_args(windows)
    assert sd * nw == D
    if var is None:
        mode = VAR_UNIT
    elif var.dim() == 1:
        mode = VAR_GLOBAL
        assert var.shape[0] == D and var.dtype == grad_out.dtype and var.is_contiguous()
    else:
        mode = VAR_FRAME
        assert var.shape == (B, T, D) and var.dtype == grad_out.dtype and var.is_contiguous()
    if out_dtype is None:
        out_dtype = torch.float32
    grad = torch.empty((B, T, D), dtype=out_dtype, device=grad_out.device)
    status = torch.empty((B * sd,), dtype=torch.int32, device=grad_out.device) if want_status else None
    rc = lib().mlpg_hip_backward(grad_out.device.index, _stream(grad_out.device), _dt(grad_out), _dt(grad), algo,
                                 _p(var), mode, _p(grad_out), _p(lengths), B, T, D, nw, pl, pu, pc,
                                 _p(grad), _p(status))
    _check(rc, "mlpg_hip_backward")
    return grad, status


def delta_features(x, windows, lengths=None):
    """Batched delta features on device tensors: x (B, T, D) -> (B, T, D * len(windows))."""
    torch = torch_mod()
    assert x.is_cuda and x.dim() == 3 and x.is_contiguous()
    B, T, D = x.shape
    nw = len(windows)
    wl, wu, wc = pack_windows(windows)
    out = torch.empty((B, T, D * nw), dtype=x.dtype, device=x.device)
    rc = lib().mlpg_hip_delta_features(x.device.index, _stream(x.device), _dt(x), _p(x), _p(lengths), B, T, D, nw,
                                       _np(wl), _np(wu), _np(wc), _p(out))
    _check(rc, "mlpg_hip_delta_features")
    return out


def _f64_3d(x):
    torch = torch_mod()
    assert x.is_cuda and x.dim() == 3
    return x.to(torch.float64).contiguous()


def modspec(x, n, ortho=False, want_phase=False):
    """Power of the n-point DFT along time of a (B, T, D) CUDA batch: (ms (B, n/2+1, D), phase (B, n/2+1, D, 2) | None)."""
    torch = torch_mod()
    x = _f64_3d(x)
    B, T, D = x.shape
    ms = torch.empty((B, n // 2 + 1, D), dtype=torch.float64, device=x.device)
    ph = torch.empty((B, n // 2 + 1, D, 2), dtype=torch.float64, device=x.device) if want_phase else None
    _check(lib().mlpg_hip_modspec(x.device.index, _stream(x.device), _p(x), B, T, D, int(n), int(bool(ortho)), _p(ms), _p(ph)),
           "mlpg_hip_modspec")
    return ms, ph


def inv_modspec(ms, phase, ortho=False):
    """irfft of sqrt(ms) * phase: ms (B, nb, D), phase (B, nb, D, 2) float64 CUDA -> (B, 2*(nb-1), D)."""
    torch = torch_mod()
    ms = _f64_3d(ms)
    phase = phase.to(torch.float64).contiguous()
    B, nb, D = ms.shape
    assert phase.shape == (B, nb, D, 2)
    n = 2 * (nb - 1)
    out = torch.empty((B, n, D), dtype=torch.float64, device=ms.device)
    _check(lib().mlpg_hip_inv_modspec(ms.device.index, _stream(ms.device), _p(ms), _p(phase), B, n, D, int(bool(ortho)), _p(out)),
           "mlpg_hip_inv_modspec")
    return out


def modspec_smoothing(x, n, limit_bin, log_domain=True, ortho=False):
    """Remove the modulation bins >= limit_bin of every column of a (B, T, D) CUDA batch; returns (B, T, D) float64."""
    torch = torch_mod()
    x = _f64_3d(x)
    B, T, D = x.shape
    out = torch.empty((B, T, D), dtype=torch.float64, device=x.device)
    _check(lib().mlpg_hip_modspec_smoothing(x.device.index, _stream(x.device), _p(x), B, T, D, int(n), int(bool(ortho)),
                                            int(limit_bin), int(bool(log_domain)), _p(out)), "mlpg_hip_modspec_smoothing")
    return out


def modspec_backward(x, grad_ms, n, ortho=False):
    """Gradient of the power spectrum w.r.t. the trajectory: x (B, T, D), grad_ms (B, n/2+1, D) -> (B, T, D)."""
    torch = torch_mod()
    x = _f64_3d(x)
    g = _f64_3d(grad_ms)
    B, T, D = x.shape
    assert g.shape == (B, n // 2 + 1, D)
    out = torch.empty((B, T, D), dtype=torch.float64, device=x.device)
    _check(lib().mlpg_hip_modspec_backward(x.device.index, _stream(x.device), _p(x), _p(g), B, T, D, int(n), int(bool(ortho)),
                                           _p(out)), "mlpg_hip_modspec_backward")
    return out


def trim_lengths(X, eps=1e-7):
    """int32 (N,) device tensor: frames left after the trailing-zero trim of each utterance."""
    torch = torch_mod()
    assert X.is_cuda and X.dim() == 3 and X.is_contiguous()
    N, T, D = X.shape
    lengths = torch.empty((N,), dtype=torch.int32, device=X.device)
    rc = lib().mlpg_hip_trim_lengths(X.device.index, _stream(X.device), _dt(X), _p(X), N, T, D, float(eps),
                                     _p(lengths))
    _check(rc, "mlpg_hip_trim_lengths")
    return lengths


DIST_L2, DIST_SCALED_L2_NP, DIST_SCALED_L1_NP, DIST_SCALED_SQL2_NP = 0, 1, 2, 3
TIE_FIRST_MIN, TIE_DIAG_LAST = 0, 1   # include/mlpg_hip.h MLPG_HIP_TIE_*


def fastdtw_l2(X, Y, lenx, leny, radius=1, dist_kind=DIST_L2, dist_scale=1.0, tie_rule=TIE_FIRST_MIN):
    """fastdtw paths for N pairs. Returns (path_i, path_j (N, Tx+Ty) int32, path_len (N,), cost (N,)).
    dist_kind / dist_scale: the local distance (include/mlpg_hip.h MLPG_HIP_DIST_*); tie_rule: MLPG_HIP_TIE_*."""
    torch = torch_mod()
    assert X.is_cuda and Y.is_cuda and X.dtype == torch.float64 and Y.dtype == torch.float64
    assert X.is_contiguous() and Y.is_contiguous() and X.dim() == 3 and Y.dim() == 3
    N, Tx, D = X.shape
    assert Y.shape[0] == N and Y.shape[2] == D
    Ty = Y.shape[1]
    dev = X.device
    path_i = torch.empty((N, Tx + Ty), dtype=torch.int32, device=dev)
    path_j = torch.empty((N, Tx + Ty), dtype=torch.int32, device=dev)
    path_len = torch.empty((N,), dtype=torch.int32, device=dev)
    cost = torch.empty((N,), dtype=torch.float64, device=dev)
    rc = lib().mlpg_hip_fastdtw(dev.index, _stream(dev), _p(X), _p(Y), _p(lenx), _p(leny), N, Tx, Ty, D,
                                int(radius), int(dist_kind), float(dist_scale), int(tie_rule), _p(path_i), _p(path_j),
                                _p(path_len), _p(cost))
    _check(rc, "mlpg_hip_fastdtw")
    return path_i, path_j, path_len, cost


def fastdtw_callable(xs, ys, radius, dist, tie_rule=TIE_FIRST_MIN, device=None):
    """fastdtw of N pairs for an ARBITRARY Python ``dist`` (what the reference hands to fastdtw, alignment.py:35-50):
    level by level from the coarsest, the local costs of each level's window cells evaluated HERE by ``dist`` -- one
    call per cell, as upstream fastdtw does -- and the DP recurrence, the back-trace and the window expansion on the GPU
    (mlpg_hip_dtw_level_windows / mlpg_hip_dtw_level_from_costs).  xs, ys: lists of (T, D) arrays (already trimmed).
    Returns numpy (path_i, path_j (N, max(tx+ty)) int32, path_len (N,), cost (N,) float64)."""
    torch = torch_mod()
    dev = require_gpu(device)
    N = len(xs)
    r = int(radius)
    # the halving pyramids, exactly upstream's __reduce_by_half (pairwise means, odd tail dropped), in float64
    px, py, Kn = [], [], []
    for x, y in zip(xs, ys):
        lx, ly = [np.asanyarray(x, dtype="float")], [np.asanyarray(y, dtype="float")]
        while not (len(lx[-1]) < r + 2 or len(ly[-1]) < r + 2):
            for l in (lx, ly):
                a = l[-1]
                h = len(a) // 2
                l.append((a[0:2 * h:2] + a[1:2 * h:2]) / 2)
        px.append(lx)
        py.append(ly)
        Kn.append(len(lx) - 1)
    tx0 = np.asarray([len(l[0]) for l in px], dtype=np.int64)
    ty0 = np.asarray([len(l[0]) for l in py], dtype=np.int64)
    row_stride = int(tx0.max())
    max_ty = int(ty0.max())
    path_stride = int((tx0 + ty0).max())
    i32 = dict(dtype=torch.int32, device=dev)
    path_i = torch.zeros((N, path_stride), **i32)
    path_j = torch.zeros((N, path_stride), **i32)
    path_len = torch.zeros((N,), **i32)
    cost = torch.zeros((N,), dtype=torch.float64, device=dev)
    row_lo = torch.zeros((N, row_stride), **i32)
    row_hi = torch.zeros((N, row_stride), **i32)
    row_off = torch.zeros((N, row_stride + 1), dtype=torch.int64, device=dev)
    st = _stream(dev)
    L = lib()
    for k in range(max(Kn), -1, -1):
        ltx = np.asarray([(int(tx0[n]) >> k) if k <= Kn[n] else 0 for n in range(N)], dtype=np.int32)
        lty = np.asarray([(int(ty0[n]) >> k) if k <= Kn[n] else 0 for n in range(N)], dtype=np.int32)
        full = np.asarray([1 if k == Kn[n] else 0 for n in range(N)], dtype=np.int32)
        d_tx, d_ty, d_full = (torch.from_numpy(a).to(dev) for a in (ltx, lty, full))
        _check(L.mlpg_hip_dtw_level_windows(dev.index, st, N, r, _p(d_tx), _p(d_ty), _p(d_full), _p(path_i), _p(path_j),
                                            _p(path_len), path_stride, _p(row_lo), _p(row_hi), _p(row_off), row_stride),
               "mlpg_hip_dtw_level_windows")
        lo, hi, off = row_lo.cpu().numpy(), row_hi.cpu().numpy(), row_off.cpu().numpy()
        ncell = np.asarray([int(off[n, ltx[n]]) if ltx[n] > 0 else 0 for n in range(N)], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(ncell)]).astype(np.int64)
        costs = np.empty(max(1, int(base[-1])), dtype=np.float64)
        for n in range(N):
            if ltx[n] <= 0:
                continue
            xk, yk = px[n][k], py[n][k]
            q = int(base[n])
            for i in range(int(ltx[n])):
                xi = xk[i]
                for j in range(int(lo[n, i]), int(hi[n, i]) + 1):
                    costs[q] = dist(xi, yk[j])          # the user's callable, once per window cell
                    q += 1
        d_costs = torch.from_numpy(costs).to(dev)
        d_base = torch.from_numpy(base[:N].copy()).to(dev)
        _check(L.mlpg_hip_dtw_level_from_costs(dev.index, st, N, int(tie_rule), _p(d_tx), _p(d_ty), _p(row_lo), _p(row_hi),
                                               _p(row_off), row_stride, max_ty, _p(d_costs), _p(d_base), int(base[-1]),
                                               _p(path_i), _p(path_j), _p(path_len), path_stride, _p(cost)),
               "mlpg_hip_dtw_level_from_costs")
        torch.cuda.current_stream(dev).synchronize()     # the level's host buffers are released only now
    pl = path_len.cpu().numpy()
    pi, pj = path_i.cpu().numpy(), path_j.cpu().numpy()
    return pi, pj, pl, cost.cpu().numpy()


def fastdtw_host(X, Y, radius=1, dist_kind=DIST_L2, dist_scale=1.0, lenx=None, leny=None, eps=1e-7, device=None,
                 tie_rule=TIE_FIRST_MIN):
    """fastdtw paths for N pairs held in numpy arrays (no framework tensor): mlpg_hip_fastdtw_host, chunked and
    overlapped with the transfers.  X (N, Tx, D), Y (N, Ty, D) float32 / float64.  Without lengths the trailing
    all-zero frames are trimmed on the device (eps as trim_zeros_frames).  Returns numpy
    (path_i, path_j (N, Tx+Ty) int32, path_len (N,), cost (N,), lenx, leny)."""
    L = lib()
    if not X.flags.c_contiguous:
        X = np.ascontiguousarray(X)
    if not Y.flags.c_contiguous:
        Y = np.ascontiguousarray(Y)
    # one dtype for the C entry point; mixed or non-float inputs are WIDENED to float64 (each array from its own
    # dtype, as the device route and the reference's fastdtw do), never narrowed
    if X.dtype != Y.dtype or X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64, copy=False)
        Y = Y.astype(np.float64, copy=False)
    assert X.ndim == 3 and Y.ndim == 3 and X.shape[0] == Y.shape[0] and X.shape[2] == Y.shape[2]
    N, Tx, D = X.shape
    Ty = Y.shape[1]
    # (the library fills every slot: the path's entries, zeros behind them -- nothing to clear here; addresses as ints: numpy's
    # `.ctypes.data_as` costs microseconds per argument, and DTWAligner.transform on one pair is a 0.3 ms call)
    path_i = np.empty((N, Tx + Ty), dtype=np.int32)
    path_j = np.empty((N, Tx + Ty), dtype=np.int32)
    small = np.zeros((3, N), dtype=np.int32)           # path_len, lenx, leny
    path_len, lx_out, ly_out = small[0], small[1], small[2]
    cost = np.zeros((N,), dtype=np.float64)
    plx = ply = None
    if lenx is not None:
        lenx = np.ascontiguousarray(lenx, dtype=np.int32)
        leny = np.ascontiguousarray(leny, dtype=np.int32)
        plx, ply = lenx.ctypes.data, leny.ctypes.data
    dt = F32 if X.dtype == np.float32 else F64
    tail = (N, Tx, Ty, D, int(radius), int(dist_kind), float(dist_scale), int(tie_rule), float(eps),
            path_i.ctypes.data, path_j.ctypes.data, path_len.ctypes.data, cost.ctypes.data, lx_out.ctypes.data, ly_out.ctypes.data)
    if isinstance(device, (list, tuple, np.ndarray)) or (isinstance(device, str) and device == "all"):
        devs = device_list(device)
        rc = L.mlpg_hip_fastdtw_host_multi(_np(devs) if len(devs) else None, len(devs), dt, X.ctypes.data, Y.ctypes.data, plx, ply, *tail)
    else:
        rc = L.mlpg_hip_fastdtw_host(current_device_index(device), dt, X.ctypes.data, Y.ctypes.data, plx, ply, *tail)
    if rc != 0:
        _check(rc, "mlpg_hip_fastdtw_host")
    return path_i, path_j, path_len, cost, lx_out, ly_out


def gather_path(src, path, path_len, Tout):
    """out[n, k] = src[n, path[n, k]] for k < path_len[n], zeros after. (N, Tout, D)."""
    torch = torch_mod()
    assert src.is_cuda and src.dim() == 3 and src.is_contiguous() and path.is_contiguous()
    N, Tsrc, D = src.shape
    out = torch.empty((N, Tout, D), dtype=src.dtype, device=src.device)
    rc = lib().mlpg_hip_gather_path(src.device.index, _stream(src.device), _dt(src), _p(src), _p(path),
                                    _p(path_len), N, Tsrc, path.shape[1], D, Tout, _p(out))
    _check(rc, "mlpg_hip_gather_path")
    return out


_MSE_WORKSPACE = {}           # (device, stream) -> [tensor, used inside a stream capture]
_MSE_WORKSPACE_RETIRED = []   # outgrown workspaces a captured graph may still replay kernels on (only those: see below)
_MSE_NEED = {}                # (B, T, D, nw, with FIR dy buffer) -> bytes (mlpg_hip_unit_mse_workspace_bytes[_t])
_MSE_FORM = {}                # (device, dtype, lengths given, B, T, D, id(packed windows)) -> (packed windows, form): settled answers only


def unit_mse_form(device, dtype, has_lengths, B, T, D, windows):
    """Which form mlpg_hip_unit_mse_step takes (mlpg_hip_unit_mse_form: the library's own decision, not a copy of it):
    2 the FIR form, 1 the one-launch kernel, 0 neither.  Answers for a PackedWindows object are remembered, except a
    "not 2" given while the stream is being captured (the tap table could not be built then; it may exist later)."""
    torch = torch_mod()
    packed = isinstance(windows, PackedWindows)
    key = (device.index, dtype, bool(has_lengths), B, T, D, id(windows)) if packed else None
    if key is not None:
        hit = _MSE_FORM.get(key)
        if hit is not None and hit[0] is windows:
            return hit[1]
    nw, pl, pu, pc, _keep = _win_args(windows)
    form = lib().mlpg_hip_unit_mse_form(device.index, _stream(device), F32 if dtype == torch.float32 else F64, 1 if has_lengths else 0,
                                        B, T, D, nw, pl, pu, pc)
    if form < 0:
        _check(form, "mlpg_hip_unit_mse_form")
    if key is not None and (form == 2 or not torch.cuda.is_current_stream_capturing()):
        if len(_MSE_FORM) > 1024:
            _MSE_FORM.clear()
        _MSE_FORM[key] = (windows, form)
    return form


def _mse_workspace(device, B, T, D, nw, fir_form):
    """The per-(device, stream) workspace of mlpg_hip_unit_mse_step.  Zeroed once (the kernel leaves its arrival counter
    zero); grown geometrically (length-bucketed or ascending batches would otherwise reallocate at almost every new
    maximum); the FIR form's dy buffer (B * T * sd floats) is only asked for by batches that take that form.  A workspace
    is never created or grown while its stream is being captured -- the allocation would come from the graph's private
    pool and its one-time zero fill would become a node of the graph, replayed behind other launches' backs: warm up one
    step of the largest shape before capturing.  An outgrown buffer is kept alive only if a stream capture has used it (a
    graph may still replay kernels on it)."""
    torch = torch_mod()
    nk = (B, T, D, nw, fir_form)
    need = _MSE_NEED.get(nk)
    if need is None:
        L = lib()
        need = int(L.mlpg_hip_unit_mse_workspace_bytes_t(B, T, D, nw) if fir_form else L.mlpg_hip_unit_mse_workspace_bytes(B, D, nw))
        if len(_MSE_NEED) > 4096:
            _MSE_NEED.clear()
        _MSE_NEED[nk] = need
    stream = _stream(device)
    key = (device.index, stream)
    ent = _MSE_WORKSPACE.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0].numel() < need:
        if capturing:
            raise HipExtensionError(
                "unit_mse_step: the workspace of this stream does not exist yet or is too small (%d bytes needed) and cannot be "
                "allocated while the stream is being captured -- run one eager step of this shape on the capturing stream first "
                "(torch.cuda.graph(g, stream=s) with the stream s the warm-up ran on: the default capture stream is torch's own)" % need)
        size = need if ent is None else max(need, ent[0].numel() * 3 // 2)
        if ent is not None and ent[1]:
            _MSE_WORKSPACE_RETIRED.append(ent[0])
        ent = [torch.zeros((size + 4095) // 4096 * 4096, dtype=torch.uint8, device=device), False]
        _MSE_WORKSPACE[key] = ent
    if capturing:
        ent[1] = True
    return ent[0], stream


def unit_mse_step(mean, target, windows, lengths=None, n_elems=None, want_y=False, want_status=False):
    """Fused unit-variance MLPG + MSE training step on device tensors (mlpg_hip_unit_mse_step): mean (B, T, D), target
    (B, T, D / nw), float32 or float64.  Returns (loss float64 0-dim tensor, grad_mean (B, T, D), y or None, status or None).
    One launch (the wave-per-system kernel: window extents <= 1, T <= 1024), or two in the FIR form (float32, no lengths, T >= 96,
    window extents <= 2, any T, a window set whose inverse decays: unit_mse_form says which)."""
    torch = torch_mod()
    assert mean.is_cuda and mean.dim() == 3 and mean.is_contiguous() and target.is_contiguous()
    B, T, D = mean.shape
    nw, pl, pu, pc, _keep = _win_args(windows)
    sd = D // nw
    assert target.shape == (B, T, sd) and target.dtype == mean.dtype and target.device == mean.device
    if lengths is not None:
        assert lengths.dtype == torch.int32 and lengths.shape == (B,) and lengths.device == mean.device
    if n_elems is None:
        n_elems = float(B * T * sd)
    grad = torch.empty_like(mean)
    y = torch.empty_like(target) if want_y else None
    loss = torch.empty((), dtype=torch.float64, device=mean.device)
    status = torch.empty((B * sd,), dtype=torch.int32, device=mean.device) if want_status else None
    # (the FIR form needs room for its dy buffer: the library says whether this call takes it)
    fir_form = mean.dtype == torch.float32 and lengths is None and T >= 96 and \
        unit_mse_form(mean.device, mean.dtype, False, B, T, D, windows) == 2
    ws, stream = _mse_workspace(mean.device, B, T, D, nw, fir_form)
    rc = lib().mlpg_hip_unit_mse_step(mean.device.index, stream, _dt(mean), _p(mean), _p(target), _p(lengths),
                                      B, T, D, nw, pl, pu, pc, float(n_elems), _p(y), _p(grad), _p(loss),
                                      _p(status), _p(ws), ws.numel())
    _check(rc, "mlpg_hip_unit_mse_step")
    return loss, grad, y, status


def stream_copy(src, dst):
    """dst[...] = src[...] by the library's plain streaming-copy kernel (measurement aid: the HBM rate a copy reaches)."""
    assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()
    nbytes = src.numel() * src.element_size()
    assert nbytes == dst.numel() * dst.element_size() and nbytes % 16 == 0
    _check(lib().mlpg_hip_stream_copy(src.device.index, _stream(src.device), _p(src), _p(dst), nbytes), "mlpg_hip_stream_copy")


def gmm_convert(x, posterior, mixture, mu_x, mu_y, A):
    """out[n] = sum_m posterior[n, m] (mu_y[m] + A[m] (x[n] - mu_x[m])) on the GPU; float64 CUDA tensors:
    x (N, D), posterior (N, M) or None, mixture int32 (N) or None, mu_x (M, D), mu_y (M, Dy), A (M, Dy, D)."""
    torch = torch_mod()
    assert x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.is_contiguous()
    N, D = x.shape
    M, Dy = mu_y.shape
    assert mu_x.shape == (M, D) and A.shape == (M, Dy, D) and A.is_contiguous()
    out = torch.empty((N, Dy), dtype=torch.float64, device=x.device)
    rc = lib().mlpg_hip_gmm_convert(x.device.index, _stream(x.device), _p(x), _p(posterior), _p(mixture), _p(mu_x),
                                    _p(mu_y), _p(A), N, D, Dy, M, _p(out))
    _check(rc, "mlpg_hip_gmm_convert")
    return out


def raise_on_status(status, sd):
    """Raise the reference's LinAlgError for the first failing (utterance, dim) system."""
    st = status.cpu().numpy().ravel()
    bad = np.flatnonzero(st)
    if bad.size:
        k = int(st[bad[0]])
        # scipy.linalg.LinAlgError is numpy.linalg.LinAlgError (linalg.pyx:79-82)
        raise np.linalg.LinAlgError("%d-th leading minor not positive definite" % k)
