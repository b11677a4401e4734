"""hipEvent_t handles through ctypes (profiling only: bench.py passes them to the library's
`phase_events` hook so that per-kernel durations are measured on the stream the kernels run on)."""
import ctypes as C

_rt = None


def _runtime():
    global _rt
    if _rt is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _rt = C.CDLL(name)
                break
            except OSError:
                continue
        if _rt is None:
            raise RuntimeError("libamdhip64.so not found")
        _rt.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _rt.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _rt.hipEventSynchronize.argtypes = [C.c_void_p]
        _rt.hipEventDestroy.argtypes = [C.c_void_p]
    return _rt


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        rc = _runtime().hipEventCreate(C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"hipEventCreate failed: {rc}")

    def synchronize(self):
        _runtime().hipEventSynchronize(self.h)

    def elapsed_ms(self, later):
        ms = C.c_float()
        rc = _runtime().hipEventElapsedTime(C.byref(ms), self.h, later.h)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed: {rc}")
        return ms.value

    def __del__(self):
        try:
            _runtime().hipEventDestroy(self.h)
        except Exception:
            pass
