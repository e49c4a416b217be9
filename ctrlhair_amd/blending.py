"""The blending step that follows the generator, on the HIP library.

`PoissonBlender(handle)` is a drop-in for the reference's `poisson_blending.poisson_blending(source, target, mask,
with_gamma)` (poisson_blending.py:29-87) and for the mask construction of hair_editor.py:297-305 (`blend_mask`): numpy /
torch uint8 in, numpy uint8 out, all arithmetic in `ch_poisson_blend` / `ch_blend_mask`.  No CPU fallback: without the
library these raise."""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib


class PoissonBlender:
    def __init__(self, handle: _lib.Handle, device: torch.device, max_iters: int = 4000, rel_tol: float = 1e-7):
        self.handle, self.device = handle, device
        self.max_iters, self.rel_tol = max_iters, rel_tol
        self.last_iters = 0
        self.last_converged = True

    def _u8(self, a, shape):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
        t = t.to(self.device).to(torch.uint8).reshape(shape).contiguous()
        return t

    def blend_mask(self, target_parsing, face_parsing) -> torch.Tensor:
        """hair_editor.py:297-305 -> res_mask_dilated uint8 [H,W] on the device (1 = generated image is kept)."""
        tp = np.asarray(target_parsing) if not isinstance(target_parsing, torch.Tensor) else target_parsing
        H, W = tp.shape[-2:]
        t, f = self._u8(target_parsing, (H, W)), self._u8(face_parsing, (H, W))
        out = torch.empty(H, W, dtype=torch.uint8, device=self.device)
        self.handle.call('ch_blend_mask', t.data_ptr(), f.data_ptr(), out.data_ptr(), H, W,
                         torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def __call__(self, source, target, mask, with_gamma=True) -> np.ndarray:
        """source, target: [H,W,3] uint8 (cv2 layout); mask [H,W] / [H,W,1], non-zero = keep source gradients, zero = keep
        the target pixel -> blended uint8 [H,W,3]."""
        H, W = int(source.shape[0]), int(source.shape[1])
        s, t = self._u8(source, (H, W, 3)), self._u8(target, (H, W, 3))
        m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(mask)))
        m = (m.to(self.device).reshape(H, W) != 0).to(torch.uint8).contiguous()
        out = torch.empty(H, W, 3, dtype=torch.uint8, device=self.device)
        iters = C.c_int(0)
        self.handle.call('ch_poisson_blend', s.data_ptr(), t.data_ptr(), m.data_ptr(), out.data_ptr(), H, W,
                         1 if with_gamma else 0, self.max_iters, float(self.rel_tol), C.byref(iters),
                         torch.cuda.current_stream(self.device).cuda_stream)
        # negated count (INT_MIN for zero iterations) = rel_tol not reached (include/ctrlhair_hip.h)
        self.last_iters = 0 if iters.value == -2 ** 31 else abs(iters.value)
        self.last_converged = iters.value >= 0
        if not self.last_converged:
            # the reference solves the system directly (poisson_blending.py:80-85): a partially converged image is not its result
            import warnings
            warnings.warn(f'Poisson blending stopped after {self.last_iters} CG iterations without reaching rel_tol='
                          f'{self.rel_tol:g}; raise PoissonBlender.max_iters', RuntimeWarning)
        return out.cpu().numpy()
