"""ctypes binding of libctrlhair_hip.so (the C ABI declared in include/ctrlhair_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.  PyTorch is
used only as the owner of device memory / streams; raw pointers cross the boundary.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libctrlhair_hip.so')

MODEL_SEAN, MODEL_SHAPE, MODEL_COLOR, MODEL_BISENET = 0, 1, 2, 3
F32, I64 = 0, 1

# every symbol include/ctrlhair_hip.h declares: name -> (restype, argtypes)
_VP, _I, _D = C.c_void_p, C.c_int, C.c_double
SYMBOLS = {
    'ch_abi_version': (_I, []),
    'ch_create': (_I, [_I, C.POINTER(_VP)]),
    'ch_destroy': (None, [_VP]),
    'ch_last_error': (C.c_char_p, [_VP]),
    'ch_load_tensor': (_I, [_VP, _I, C.c_char_p, _VP, _I, C.POINTER(C.c_int64), _I]),
    'ch_set_option': (_I, [_VP, C.c_char_p, _I]),
    'ch_finalize': (_I, [_VP, _I, _I, _I]),
    'ch_sean_noise_floats': (C.c_size_t, [_VP, _I]),
    'ch_sean_generate': (_I, [_VP, _VP, _VP, _VP, C.c_uint64, _VP, _I, _I, _VP]),
    'ch_sean_draw_noise': (_I, [_VP, C.c_uint64, _VP, _I, _I, _VP]),
    'ch_sean_encode': (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    'ch_sean_encode_features': (_I, [_VP, _VP, _I, _I, _VP]),
    'ch_sean_encode_regions': (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    'ch_color_generate': (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    'ch_color_encode': (_I, [_VP, _VP, _VP, _I, _VP]),
    'ch_color_predict': (_I, [_VP, _VP, _VP, _I, _VP]),
    'ch_shape_encode': (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    'ch_shape_decode': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    'ch_shape_combine': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP]),
    'ch_bisenet_parse': (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    'ch_blend_mask': (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    'ch_poisson_blend': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _D, C.POINTER(_I), _VP]),
    'ch_sean_set_tap': (_I, [_VP, C.c_char_p, _VP]),
    'ch_sean_scale_report': (_I, [_VP, C.POINTER(C.c_float), _I]),
    'ch_sean_debug_read': (_I, [_VP, _VP, C.c_size_t]),
    'ch_mfma_peak': (_I, [_VP, _I, _I, C.POINTER(_D)]),
    'ch_profile_enable': (_I, [_VP, _I]),
    'ch_profile_read': (_I, [_VP, _I, C.POINTER(_I), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    'ch_profile_read_ex': (_I, [_VP, _I, C.POINTER(_I), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
}

_lib = None


def load():
    """dlopen the library and declare prototypes.  Raises if it has not been built (__graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                               f'g.build()"` or `make -C ctrlhair_amd/csrc` (no CPU fallback exists)')
        # PyTorch ships its own libamdhip64; the callers of this module hand us torch device pointers and streams, so both
        # must live in ONE HIP runtime.  Importing torch first makes the dynamic loader resolve our DT_NEEDED libamdhip64
        # to the copy torch already mapped (loading /opt/rocm's first leaves torch with "No HIP GPUs are available").
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class Handle:
    """Owns one ch_handle (one device).  Not thread-safe, like the reference's Backend."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _VP()
        rc = self.lib.ch_create(device, C.byref(h))
        self._h = h
        if rc != 0:
            msg = self.lib.ch_last_error(h).decode() if h else 'no HIP device / ch_create failed'
            if h:
                self.lib.ch_destroy(h)
                self._h = None
            raise RuntimeError(f'ch_create(device={device}) failed: {msg}')
        self.device = device

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed ({rc}): {self.lib.ch_last_error(self._h).decode()}')

    def load_tensor(self, model: int, name: str, arr):
        import numpy as np
        a = np.ascontiguousarray(arr)
        if a.dtype == np.float32:
            dt = F32
        elif a.dtype == np.int64:
            dt = I64
        else:
            raise TypeError(f'{name}: unsupported dtype {a.dtype}')
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        self._check(self.lib.ch_load_tensor(self._h, model, name.encode(), a.ctypes.data_as(_VP), dt, shape, a.ndim),
                    f'ch_load_tensor({name})')

    def set_option(self, key: str, value: int):
        self._check(self.lib.ch_set_option(self._h, key.encode(), int(value)), f'ch_set_option({key})')

    def finalize(self, model: int, max_batch: int, max_size: int):
        self._check(self.lib.ch_finalize(self._h, model, max_batch, max_size), 'ch_finalize')

    def sean_noise_floats(self, S: int) -> int:
        return int(self.lib.ch_sean_noise_floats(self._h, S))

    def sean_generate(self, labels_ptr, codes_ptr, noise_ptr, seed, out_ptr, B, S, stream_ptr):
        self._check(self.lib.ch_sean_generate(self._h, labels_ptr, codes_ptr, noise_ptr, seed, out_ptr, B, S,
                                              stream_ptr), 'ch_sean_generate')

    def sean_encode(self, img_ptr, labels_ptr, codes_ptr, B, S, stream_ptr):
        self._check(self.lib.ch_sean_encode(self._h, img_ptr, labels_ptr, codes_ptr, B, S, stream_ptr),
                    'ch_sean_encode')

    def call(self, fn: str, *args):
        """Generic checked call of a ch_* entry point taking (handle, *args)."""
        self._check(getattr(self.lib, fn)(self._h, *args), fn)

    def sean_set_tap(self, name: str, ptr):
        self._check(self.lib.ch_sean_set_tap(self._h, name.encode(), ptr), 'ch_sean_set_tap')

    def sean_scale_report(self):
        """f16x3 / f16 paths: maxima of |value * 8| recorded by the producers of the dynamically scaled activation
        tensors during the last generate chunk, as {'ace': float[18], 'style': float[18]} (0 = not written).  A value
        outside [0.5, 65504] means that tensor was rewritten with a corrected scale (csrc/sh16.h).  Synchronises."""
        import numpy as np
        buf = (C.c_float * 36)()
        self._check(self.lib.ch_sean_scale_report(self._h, buf, 36), 'ch_sean_scale_report')
        a = np.array(buf[:], dtype=np.float32)
        return {'ace': a[0::2].copy(), 'style': a[1::2].copy()}

    def mfma_peak(self, kind: int, ms_target: int = 30) -> float:
        """TFLOP/s this device sustains on an MFMA-only loop (kind 0: f32 32x32x2, 1: f16 32x32x16).  Synchronises."""
        t = _D()
        self._check(self.lib.ch_mfma_peak(self._h, kind, ms_target, C.byref(t)), 'ch_mfma_peak')
        return t.value

    def profile_enable(self, on: bool):
        self._check(self.lib.ch_profile_enable(self._h, int(on)), 'ch_profile_enable')

    def profile_read(self, kind: int = -1):
        """'flops': the dense evaluation of the recorded layers; 'flops_executed': what the matrix cores ran (smaller for
        ACE launches on the exact SPADE-interior reduction, csrc/ace_sparse.h)."""
        n, ms, fl, fx, by = _I(), _D(), _D(), _D(), _D()
        self._check(self.lib.ch_profile_read_ex(self._h, kind, C.byref(n), C.byref(ms), C.byref(fl), C.byref(fx), C.byref(by)),
                    'ch_profile_read_ex')
        return {'launches': n.value, 'ms': ms.value, 'flops': fl.value, 'flops_executed': fx.value, 'bytes': by.value}

    def close(self):
        if getattr(self, '_h', None):
            self.lib.ch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
