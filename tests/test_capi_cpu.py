"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every
symbol include/mlpg_hip.h declares, argument validation works without a GPU,
and the host-side mirrors of the reference helpers behave like the reference."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from nnmnkwii_amd.csrc import build as hip_build
    hip_build.build()
    from nnmnkwii_amd import _hip
    return _hip.lib()


def test_exports_match_header(L):
    from nnmnkwii_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "mlpg_hip.h")).read()
    declared = set(re.findall(r"\b(mlpg_hip_\w+)\s*\(", hdr))
    assert declared == set(_hip.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.mlpg_hip_abi_version() == _hip.ABI_VERSION == 14


def test_argument_validation_without_gpu(L):
    wl = np.array([0, 1], dtype=np.int32)
    wu = np.array([0, 1], dtype=np.int32)
    wc = np.array([1.0, -0.5, 0.0, 0.5])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    fake = ctypes.c_void_p(64)
    # D not a multiple of num_windows
    rc = L.mlpg_hip_forward(0, None, 1, 0, fake, fake, 0, None, 1, 4, 5, 2, p(wl), p(wu), p(wc), fake, None)
    assert rc == -1 and b"multiple" in L.mlpg_hip_last_error()
    # bad dtype
    rc = L.mlpg_hip_forward(0, None, 7, 0, fake, fake, 0, None, 1, 4, 4, 2, p(wl), p(wu), p(wc), fake, None)
    assert rc == -1
    # window extent too large
    wl2 = np.array([0, 9], dtype=np.int32)
    rc = L.mlpg_hip_forward(0, None, 1, 0, fake, fake, 0, None, 1, 4, 4, 2, p(wl2), p(wu), p(wc), fake, None)
    assert rc == -1 and b"extents" in L.mlpg_hip_last_error()
    # empty batch is a no-op
    rc = L.mlpg_hip_forward(0, None, 1, 0, None, fake, 0, None, 0, 4, 4, 2, p(wl), p(wu), p(wc), None, None)
    assert rc == 0
    # multi-stream entry: a stream that does not fit the row, and an empty table
    from nnmnkwii_amd._hip import StreamDesc
    tab = (StreamDesc * 1)(StreamDesc(4, 0, 1, 2, 0))
    rc = L.mlpg_hip_forward_streams(0, None, 1, 0, fake, fake, 0, 5, None, 1, 4, 1, ctypes.addressof(tab), 2,
                                    p(wl), p(wu), p(wc), fake, 1, None)
    assert rc == -1 and b"does not fit" in L.mlpg_hip_last_error()
    rc = L.mlpg_hip_forward_streams(0, None, 1, 0, fake, fake, 0, 5, None, 1, 4, 0, None, 0, None, None, None, fake, 1, None)
    assert rc == 0
    # modulation spectrum: DFT length >= 2 and >= T
    rc = L.mlpg_hip_modspec(0, None, fake, 1, 1, 2, 1, 0, fake, None)
    assert rc == -1 and b"at least 2" in L.mlpg_hip_last_error()
    rc = L.mlpg_hip_modspec_smoothing(0, None, fake, 1, 100, 2, 64, 0, 10, 1, fake)
    assert rc == -1 and b"time length" in L.mlpg_hip_last_error()
    rc = L.mlpg_hip_fastdtw_l2(0, None, fake, fake, fake, fake, 1, 4, 4, 2, 0, fake, fake, fake, fake)
    assert rc == -1
    assert L.mlpg_hip_device_count() >= 0


def test_unit_mse_form_validates_without_gpu(L):
    wl = np.array([0, 1], dtype=np.int32)
    wu = np.array([0, 1], dtype=np.int32)
    wc = np.array([1.0, -0.5, 0.0, 0.5])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    assert L.mlpg_hip_unit_mse_form(0, None, 0, 0, 2, 100, 5, 2, p(wl), p(wu), p(wc)) == -1      # D not a multiple of nw
    assert b"multiple" in L.mlpg_hip_last_error()
    assert L.mlpg_hip_unit_mse_form(0, None, 9, 0, 2, 100, 4, 2, p(wl), p(wu), p(wc)) == -1      # bad dtype
    assert L.mlpg_hip_unit_mse_form(0, None, 0, 0, 0, 100, 4, 2, p(wl), p(wu), p(wc)) == 1       # an empty batch: the step answers it
    assert L.mlpg_hip_launch_count(10) >= 0 and L.mlpg_hip_launch_count(11) >= 0 and L.mlpg_hip_launch_count(12) == -1


def test_backward_host_validates_without_gpu(L):
    """mlpg_hip_backward_host (ABI 14: the literal paramgen.mlpg_grad call on host memory): argument errors and empty
    batches are answered before any device is touched."""
    wl = np.array([0, 1], dtype=np.int32)
    wu = np.array([0, 1], dtype=np.int32)
    wc = np.array([1.0, -0.5, 0.0, 0.5])
    go = np.zeros((1, 4, 2))
    var = np.ones((1, 4, 4))
    grad = np.zeros((1, 4, 4), dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    args = (p(wl), p(wu), p(wc), p(grad), None)
    assert L.mlpg_hip_backward_host(0, 1, 0, 0, p(var), 0, p(go), None, 1, 4, 5, 2, *args) == -1       # D not a multiple of nw
    assert b"bad sizes" in L.mlpg_hip_last_error()
    assert L.mlpg_hip_backward_host(0, 7, 0, 0, p(var), 0, p(go), None, 1, 4, 4, 2, *args) == -1       # bad dtype
    assert L.mlpg_hip_backward_host(0, 1, 0, 0, None, 0, p(go), None, 1, 4, 4, 2, *args) == -1         # per-frame variances, NULL
    assert L.mlpg_hip_backward_host(-1, 1, 0, 0, p(var), 0, p(go), None, 1, 4, 4, 2, *args) == -1       # bad device
    assert L.mlpg_hip_backward_host(0, 1, 0, 0, p(var), 0, p(go), None, 0, 4, 4, 2, *args) == 0        # an empty batch
    assert L.mlpg_hip_backward_host(0, 1, 0, 0, p(var), 0, p(go), None, 1, 0, 4, 2, *args) == 0        # no frames


def test_host_copy_pool_copies_every_byte_and_survives_shutdown(L):
    """mlpg_hip_host_copy: the staging copy of the short host path (calling thread + helper threads, 64 KB slices claimed
    through one atomic word) -- sizes around the slice boundaries, back-to-back jobs (helpers awake), jobs behind a pause
    (helpers asleep), from two Python threads, and again after mlpg_hip_shutdown joined the helpers."""
    import threading
    import time
    rng = np.random.RandomState(0)
    src = rng.randint(0, 256, size=(9 << 20) + 13, dtype=np.uint8)

    def check(n, off=0):
        dst = np.full(n + 64, 0xA5, dtype=np.uint8)
        assert L.mlpg_hip_host_copy(dst.ctypes.data + 32, src.ctypes.data + off, n) == 0
        assert np.array_equal(dst[32:32 + n], src[off:off + n])
        assert (dst[:32] == 0xA5).all() and (dst[32 + n:] == 0xA5).all()       # nothing outside the range is touched

    for n in (0, 1, 65535, 65536, 65537, 131071, 131072, 131073, 1440000, 2880000, (9 << 20) + 13):
        check(n)
    for k in range(200):                         # back to back: the helpers are spinning
        check(int(rng.randint(1, 3 << 20)), int(rng.randint(0, 1 << 20)))
    for _ in range(3):                           # behind a pause: the helpers sleep and are woken
        time.sleep(0.01)
        check(1440000, 7)
    errs = []

    def run(seed):
        r = np.random.RandomState(seed)
        try:
            for _ in range(60):
                check(int(r.randint(1, 2 << 20)), int(r.randint(0, 1 << 20)))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(s,)) for s in (1, 2, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    L.mlpg_hip_shutdown()                        # joins the helpers; the next copy brings them up again
    check(2880000)
    L.mlpg_hip_shutdown()
    assert L.mlpg_hip_host_copy(None, None, 0) == 0 and L.mlpg_hip_host_copy(None, src.ctypes.data, 8) == -1


def test_cached_windows_follow_the_list():
    from nnmnkwii_amd import _hip
    w = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5]))]
    a = _hip.cached_windows(w)
    assert _hip.cached_windows(w) is a and a[3] == 2 and a[2].tolist() == [1.0, -0.5, 0.0, 0.5]
    w[1][2][0] = -0.25                                   # edited in place: packed afresh
    b = _hip.cached_windows(w)
    assert b is not a and b[2].tolist() == [1.0, -0.25, 0.0, 0.5]
    w.append((1, 1, np.array([1.0, -2.0, 1.0])))
    c = _hip.cached_windows(w)
    assert c[3] == 3 and _hip.cached_windows(w) is c
    assert _hip.cached_windows(c) is c                   # already packed
    wl = [(0, 0, [1.0]), (1, 0, (-1.0, 1.0))]            # plain sequences as coefficients
    d = _hip.cached_windows(wl)
    assert d[2].tolist() == [1.0, -1.0, 1.0] and _hip.cached_windows(wl) is d
    wl[1] = (1, 0, (-2.0, 2.0))
    assert _hip.cached_windows(wl)[2].tolist() == [1.0, -2.0, 2.0]
    with pytest.raises(AssertionError):
        _hip.cached_windows([(1, 1, np.array([1.0]))])


def test_current_device_index_survives_a_half_imported_torch(monkeypatch):
    """A numpy entry point called while another thread is inside `import torch` sees a module object without attributes in
    sys.modules (found by tools/dbg/lit_threads_soak.py): the device index falls back to 0 instead of raising."""
    import sys
    import types
    from nnmnkwii_amd import _hip
    monkeypatch.setitem(sys.modules, "torch", types.ModuleType("torch"))
    assert _hip.current_device_index() == 0
    assert _hip.current_device_index("cuda:3") == 3 and _hip.current_device_index(2) == 2


def test_pack_windows():
    from nnmnkwii_amd import _hip
    wl, wu, wc = _hip.pack_windows(WINDOW_SETS["wide3"])
    assert wl.tolist() == [0, 2, 2] and wu.tolist() == [0, 2, 2] and wc.shape == (11,)
    with pytest.raises(AssertionError):
        _hip.pack_windows([(1, 1, np.array([1.0]))])


def test_win_mats_and_full_window_mat():
    # reference: tests/test_paramgen.py:62-79
    from nnmnkwii_amd import paramgen as G
    for wname, windows in WINDOW_SETS.items():
        for T in (1, 2, 5, 10):
            win_mats = G.build_win_mats(windows, T)
            fulls = [w.full() for w in win_mats]
            for (l, u, c), w, f in zip(windows, win_mats, fulls):
                assert (w.l, w.u, w.transposed) == (l, u, True)
                assert np.array_equal(f, O.window_matrix(l, u, c, T))
                assert np.array_equal(w.T.full(), f.T)
            assert np.array_equal(G.full_window_mat(win_mats, T), np.vstack(fulls))


def test_reshape_means():
    # reference: tests/test_paramgen.py:98-110
    from nnmnkwii_amd import paramgen as G
    T, sd = 10, 2
    for windows in WINDOW_SETS.values():
        means = np.random.RandomState(0).rand(T, sd * len(windows))
        r = G.reshape_means(means, sd)
        assert r.shape == (T * len(windows), sd)
        assert G.reshape_means(r, sd) is r or len(windows) == 1
        assert np.array_equal(r, O.reshape_means(means, sd))


def test_no_cpu_fallback():
    """Without a GPU the product raises instead of silently computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nnmnkwii_amd import HipExtensionError
    from nnmnkwii_amd import paramgen as G
    with pytest.raises(HipExtensionError):
        G.mlpg(np.zeros((4, 3)), np.ones((4, 3)), WINDOW_SETS["std3"])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nnmnkwii_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
                assert "liboracle" not in src, fn


def test_dtw_custom_dist_is_evaluated_on_the_host_and_needs_the_gpu_for_the_rest():
    from nnmnkwii_amd import HipExtensionError
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner, _resolve_dist
    assert _resolve_dist(lambda x, y: 1.0 - float(x @ y)) is None
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    a = DTWAligner(dist=lambda x, y: 1.0 - float(x @ y))
    with pytest.raises(HipExtensionError):       # no CPU fallback for the DP either
        a.transform((np.ones((1, 3, 2)), np.ones((1, 3, 2))))
    d = DTWAligner()
    assert d.radius == 1 and d.verbose == 0 and callable(d.dist) and d.tie_rule == "first"


def test_dtw_broken_dist_warns_instead_of_falling_back_silently():
    """a `dist` that raises on the probe frames used to be swallowed (and the alignment silently went the
    per-cell host route); now it says so, once per process."""
    import warnings
    from nnmnkwii_amd.preprocessing import alignment as A
    A._warned.discard("probe")

    def broken(x, y):
        raise ValueError("typo")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert A._resolve_dist(broken) is None
        assert A._resolve_dist(broken) is None
    assert len([x for x in w if issubclass(x.category, RuntimeWarning) and "probe frames" in str(x.message)]) == 1


def test_compat_install_provides_the_reference_names():
    """nnmnkwii_amd.compat: user code written against ``nnmnkwii`` resolves to the HIP path (no GPU needed to import)."""
    import sys
    import nnmnkwii_amd.compat as compat
    from nnmnkwii_amd import paramgen as G
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    assert "nnmnkwii" not in sys.modules
    compat.install()
    try:
        from nnmnkwii.paramgen import mlpg, unit_variance_mlpg_matrix   # noqa: F401
        from nnmnkwii.preprocessing.alignment import DTWAligner as D2, IterativeDTWAligner   # noqa: F401
        from nnmnkwii.baseline.gmm import MLPG   # noqa: F401
        from nnmnkwii.autograd import unit_variance_mlpg   # noqa: F401
        from nnmnkwii.util import apply_each2d_padded   # noqa: F401
        import nnmnkwii.preprocessing as P
        assert mlpg is G.mlpg and D2 is DTWAligner and P.modspec_smoothing is not None
        compat.install()          # idempotent
    finally:
        compat.uninstall()
    assert "nnmnkwii" not in sys.modules and "nnmnkwii.paramgen" not in sys.modules


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/mlpg_hip.h is what a cgo / JNI / ctypes binding reads: it must compile as C (no C++-isms), and a C
    program must link against the library and get its ABI version -- no GPU needed for that."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    from nnmnkwii_amd import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = _hip.SO_PATH
    if not os.path.exists(so):
        pytest.skip("library not built")
    src = tmp_path / "demo.c"
    src.write_text(
        '#include <stdio.h>\n#include "mlpg_hip.h"\n'
        "int main(void) {\n"
        "  mlpg_hip_stream_t s; s.in_col = 0; s.out_col = 0; s.static_dim = 1; s.num_windows = 0; s.win_first = 0; (void)s;\n"
        '  printf("%d %d %d\\n", mlpg_hip_abi_version(), MLPG_HIP_ALGO_CONST, MLPG_HIP_DIST_SCALED_SQL2_NP);\n'
        "  return 0;\n}\n")
    exe = tmp_path / "demo"
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), so,
           "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0, r.stdout
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=env)
    assert out.returncode == 0, out.stdout
    assert out.stdout.split() == ["14", "5", "3"], out.stdout


def test_host_chunk_plan_deals_round_robin_and_covers_the_batch():
    """mlpg_hip_host_chunk_plan: the dealing logic of the host-memory entry points over a device list (pure host code).
    Chunks tile the batch in order without gaps; chunk c runs on list entry c % n, stream slot (c // n) % 2; at least
    four chunks per device when the batch allows; never an empty chunk."""
    from nnmnkwii_amd import _hip
    for n_items, target, ndev in [(256, 10, 1), (256, 10, 8), (37, 10, 1), (37, 1000, 3), (1, 5, 8), (1024, 90, 8), (5, 2, 4),
                                  (0, 4, 2)]:
        entry, slot, first, count = _hip.host_chunk_plan(n_items, target, ndev)
        assert count.sum() == n_items and (count > 0).all()
        assert (first == np.concatenate([[0], np.cumsum(count)[:-1]])).all()
        c = np.arange(len(entry))
        assert (entry == c % ndev).all() and (slot == (c // ndev) % 2).all()
        if len(count):
            assert count.max() <= target
            assert count[0] == max(1, min(target, -(-n_items // (4 * ndev))))      # about four chunks per device
            assert (count[:-1] == count[0]).all()
    # one device: exactly the two-slot alternation of the single-device call
    entry, slot, first, count = _hip.host_chunk_plan(37, 10, 1)
    assert entry.tolist() == [0, 0, 0, 0] and slot.tolist() == [0, 1, 0, 1] and count.tolist() == [10, 10, 10, 7]
    with pytest.raises(_hip.HipExtensionError):
        _hip.host_chunk_plan(10, 0, 1)
    assert _hip.device_list("all").size == 0 and _hip.device_list([1, 1, 0]).tolist() == [1, 1, 0]
    assert _hip.device_list(3).tolist() == [3] and _hip.device_list("cuda:2").tolist() == [2]


def test_one_hip_runtime_whichever_is_loaded_first():
    """Loading libmlpg_hip.so BEFORE torch must not bring a second HIP runtime into the process (the PyTorch-ROCm wheel ships its
    own copy; with two of them torch.cuda.is_available() turns False): checked in a fresh interpreter, library first."""
    code = r"""
import sys
sys.path.insert(0, %r)
from nnmnkwii_amd import _hip
_hip.lib()
import torch
libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))
print(len(libs), libs)
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_header_constants_match_the_python_binding():
    """Every MLPG_HIP_* integer constant of include/mlpg_hip.h that the ctypes binding mirrors has the same value there."""
    from nnmnkwii_amd import _hip
    src = open(os.path.join(ROOT, "include", "mlpg_hip.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define\s+MLPG_HIP_([A-Z0-9_]+)\s+\(?(-?\d+)\)?", src, re.M)}
    pairs = {"ALGO_AUTO": _hip.ALGO_AUTO, "ALGO_GENERIC": _hip.ALGO_GENERIC, "ALGO_WAVE": _hip.ALGO_WAVE, "ALGO_STRIP": _hip.ALGO_STRIP,
             "ALGO_PIPE": _hip.ALGO_PIPE, "ALGO_CONST": _hip.ALGO_CONST, "ALGO_CHUNK": _hip.ALGO_CHUNK, "ALGO_FIR": _hip.ALGO_FIR,
             "F32": _hip.F32, "F64": _hip.F64, "VAR_FRAME": _hip.VAR_FRAME, "VAR_GLOBAL": _hip.VAR_GLOBAL, "VAR_UNIT": _hip.VAR_UNIT,
             "DIST_L2": _hip.DIST_L2, "DIST_SCALED_L2_NP": _hip.DIST_SCALED_L2_NP, "DIST_SCALED_L1_NP": _hip.DIST_SCALED_L1_NP,
             "DIST_SCALED_SQL2_NP": _hip.DIST_SCALED_SQL2_NP, "TIE_FIRST_MIN": _hip.TIE_FIRST_MIN, "TIE_DIAG_LAST": _hip.TIE_DIAG_LAST}
    for name, val in pairs.items():
        assert defs.get(name) == val, (name, defs.get(name), val)
