"""Loader for libsarlacc_amd.so (HIP kernels + C ABI, include/sarlacc_amd.h).

There is deliberately no fallback: if the shared library is missing or no HIP
device is usable, every compute call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SARLACC_LIB_PATH") or os.path.join(_HERE, "libsarlacc_amd.so")   # override: experiment builds
_lib = None


class SarlaccError(RuntimeError):
    """Error raised by the native library; the message is the reference's own
    wherever the reference would have thrown (src/utils.cpp, src/reference_align.cpp ...)."""


def _preload_shared_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two
    copies of the HIP runtime in one process cannot both own the GPU ("No HIP GPUs are
    available" in whichever initialises second), so when torch is installed its copy is loaded
    first and libsarlacc_amd.so binds to it by SONAME -- whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _try_build():
    """Build the in-tree library when it is missing and hipcc is available (fresh checkout)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if hipcc is None:
        return
    try:
        subprocess.run(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc"), "HIPCC=" + hipcc],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    except subprocess.CalledProcessError as e:
        tail = (e.stdout or b"").decode(errors="replace")[-2000:]
        raise ImportError("sarlacc_amd: building %s failed (make exit %d):\n%s" % (LIB_PATH, e.returncode, tail))
    except OSError as e:
        raise ImportError("sarlacc_amd: cannot run make to build %s: %s" % (LIB_PATH, e))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            _try_build()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "sarlacc_amd: %s not found -- build it with `make -C sarlacc_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        _preload_shared_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.sarlacc_last_error.restype = C.c_char_p
        _lib.sarlacc_last_kernel_ms.restype = C.c_double
        _lib.sarlacc_stage_ms.restype = C.c_double
        _lib.sarlacc_stage_count.restype = C.c_double
        _lib.sarlacc_release_umi_workspace.restype = C.c_int64
        _lib.sarlacc_workspace_report.restype = C.c_int64
    return _lib


def check(rc):
    if rc:
        raise SarlaccError(lib().sarlacc_last_error().decode())


def ptr(a):
    """numpy array / None -> void*"""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class _HostBlock:
    """A page-locked block of sarlacc_host_alloc behind the array interface; goes back to the library's pool with its last view."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().sarlacc_host_alloc(C.byref(p), C.c_int64(nbytes)))
        self.address = p.value
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.address, False), "version": 3}

    def __del__(self):
        try:
            if self.address:
                lib().sarlacc_host_free(C.c_void_p(self.address))
                self.address = 0
        except Exception:
            pass


_DEBUG_ZERO = os.environ.get("SARLACC_DEBUG_ZERO_HOST") == "1"


def host_array(count, dtype):
    """Uninitialised numpy array for a result the device writes in full: page-locked from 1 MB on (sarlacc_host_alloc: the
    download is one DMA transfer instead of a staged copy), ordinary memory below that.  Contents beyond what the C call
    wrote are UNDEFINED (a pooled block keeps what its last user left): callers read only the range the call reports
    (offsets[-1] of a string set, the returned count); SARLACC_DEBUG_ZERO_HOST=1 zero-fills for debugging."""
    dt = np.dtype(dtype)
    nbytes = int(count) * dt.itemsize
    if nbytes < (1 << 20):
        return np.zeros(int(count), dt) if _DEBUG_ZERO else np.empty(int(count), dt)
    arr = np.asarray(_HostBlock(nbytes))[:nbytes].view(dt)
    if _DEBUG_ZERO:
        arr[:] = 0
    return arr


def device_count():
    return int(lib().sarlacc_device_count())


def set_device(device):
    check(lib().sarlacc_set_device(int(device)))


def release_umi_workspace():
    """sarlacc_release_umi_workspace: gives the umi_group stage's cached device buffers back (bytes freed)."""
    return int(lib().sarlacc_release_umi_workspace())


def workspace_report(top=12):
    """sarlacc_workspace_report: (total bytes, [(name, bytes), ...] of the `top` largest cached device buffers)."""
    buf = C.create_string_buffer(1 << 16)
    total = int(lib().sarlacc_workspace_report(buf, C.c_int64(len(buf))))
    rows = [ln.rsplit(" ", 1) for ln in buf.value.decode().splitlines() if ln]
    return total, [(n, int(b)) for n, b in rows[:top]]


def stage_ms(name):
    """Milliseconds of the named kernel group in the last call that ran it (HIP events), <0 if none."""
    return float(lib().sarlacc_stage_ms(name.encode()))


def stage_count(name):
    """Work counter (cells, pairs ...) recorded by the last call that set it, <0 if unset."""
    return float(lib().sarlacc_stage_count(name.encode()))


def last_kernel_ms():
    return float(lib().sarlacc_last_kernel_ms())
