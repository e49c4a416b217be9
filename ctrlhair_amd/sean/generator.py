"""Host-side wrapper of the HIP SEAN generator: the object that stands where the reference keeps
``Pix2PixModel.netG`` (sean_codes/models/pix2pix_model.py:103-113) for the 'UI_mode' path.

    gen = SeanGenerator(device=0).load_state_dict(sd, max_batch=16, max_size=512)
    img = gen.generate(labels_u8[B,S,S], codes[B,19,512], noise=None|[B,NF])   # -> cuda float32 [B,3,S,S]
"""
from typing import Dict, Optional

import numpy as np
import torch

from .. import lib as _lib


class SeanGenerator:
    def __init__(self, device: int = 0, f16x3=False, options: Optional[Dict[str, int]] = None):
        """f16x3: False/0 = exact-f32 MFMA convs (v_mfma_f32_32x32x2_f32); True/1 = 3-term split-operand f16 MFMA convs with
        f32 accumulation (conv_sh16.h): f32-class results (max |delta| vs the exact path 1.5e-5), ~3x faster; 2 = single-term
        f16 operands with f32 accumulation (reduced precision, tolerance 5e-2 -- BASELINE.json configs[4])."""
        self.f16x3 = f16x3
        self.options = dict(options or {})          # extra ch_set_option(key, value) pairs applied before ch_finalize
        self.device_index = device
        self.device = torch.device('cuda', device)
        self.handle = _lib.Handle(device)
        self.max_batch = self.max_size = 0

    def load_state_dict(self, sd: Dict[str, object], max_batch: int = 16, max_size: int = 512):
        """sd: reference-keyed state dict (torch tensors or numpy arrays), e.g. torch.load('latest_net_G.pth')
        (util/util.py:202-208) or ctrlhair_amd.procedural.sean_state_dict()."""
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if a.dtype not in (np.float32, np.int64):
                a = a.astype(np.float32)
            self.handle.load_tensor(_lib.MODEL_SEAN, k, a)
        self.handle.set_option('sean.f16x3', int(self.f16x3))
        for k, v in self.options.items():
            self.handle.set_option(k, int(v))
        self.handle.finalize(_lib.MODEL_SEAN, max_batch, max_size)
        self.max_batch, self.max_size = max_batch, max_size
        return self

    def noise_floats(self, S: int) -> int:
        return self.handle.sean_noise_floats(S)

    def generate(self, labels: torch.Tensor, codes: torch.Tensor, noise: Optional[torch.Tensor] = None,
                 seed: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert labels.is_cuda and labels.dtype == torch.uint8 and labels.dim() == 3, 'labels: cuda uint8 [B,S,S]'
        B, S = labels.shape[0], labels.shape[-1]
        assert labels.shape[1] == S
        assert codes.is_cuda and codes.dtype == torch.float32 and tuple(codes.shape) == (B, 19, 512)
        labels, codes = labels.contiguous(), codes.contiguous()
        nptr = None
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and tuple(noise.shape) == (B, self.noise_floats(S))
            noise = noise.contiguous()
            nptr = noise.data_ptr()
        if out is None:
            out = torch.empty(B, 3, S, S, dtype=torch.float32, device=labels.device)
        stream = torch.cuda.current_stream(labels.device).cuda_stream
        self.handle.sean_generate(labels.data_ptr(), codes.data_ptr(), nptr, seed, out.data_ptr(), B, S, stream)
        return out

    def draw_noise(self, B: int, S: int, seed: int = 0) -> torch.Tensor:
        """The planes `generate(..., noise=None, seed=seed)` uses, as a tensor [B, noise_floats(S)]."""
        out = torch.empty(B, self.noise_floats(S), dtype=torch.float32, device=self.device)
        self.handle.call('ch_sean_draw_noise', seed, out.data_ptr(), B, S, torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def capture(self, labels: torch.Tensor, codes: torch.Tensor, noise: Optional[torch.Tensor] = None, seed: int = 0):
        """hipGraph capture of one generate() call at fixed shapes.  Returns (graph, out): refill `labels` / `codes` / `noise`
        IN PLACE, then graph.replay() and read `out`.  The library allocates nothing and never synchronises inside generate,
        so the stock torch.cuda.CUDAGraph capture applies (device-drawn noise keeps the seed of the capture; the library's
        internal side stream joins the capture through its fork / join events).  Measured (tools/lat_b1.py): a replay is NOT
        faster than eager launches -- the ~400 kernels of a render run back to back either way, and the graph executor
        serialises the run-ahead side stream (3.4 ms vs 2.65 ms at 256x256).  Kept for callers that want one submission."""
        out = torch.empty(labels.shape[0], 3, labels.shape[-1], labels.shape[-1], dtype=torch.float32, device=labels.device)
        side = torch.cuda.Stream(labels.device)
        side.wait_stream(torch.cuda.current_stream(labels.device))
        with torch.cuda.stream(side):
            self.generate(labels, codes, noise, seed=seed, out=out)        # warm-up outside the capture (lazy kernel attributes)
        torch.cuda.current_stream(labels.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.generate(labels, codes, noise, seed=seed, out=out)
        return g, out

    def encode(self, img: torch.Tensor, labels: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Zencoder: img cuda float32 [B,3,S,S] in [-1,1], labels cuda uint8 [B,S,S] -> codes [B,19,512]
        (Pix2PixModel.forward(mode='style_code'), pix2pix_model.py:69-72)."""
        assert img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3
        B, S = img.shape[0], img.shape[-1]
        assert labels.is_cuda and labels.dtype == torch.uint8 and tuple(labels.shape) == (B, S, S)
        img, labels = img.contiguous(), labels.contiguous()
        if out is None:
            out = torch.empty(B, 19, 512, dtype=torch.float32, device=img.device)
        stream = torch.cuda.current_stream(img.device).cuda_stream
        self.handle.sean_encode(img.data_ptr(), labels.data_ptr(), out.data_ptr(), B, S, stream)
        return out

    def encode_features(self, img: torch.Tensor) -> None:
        """First half of encode(): the Zencoder's convolutions (B <= max_batch); the feature map stays in the library's
        workspace until encode_regions().  The label map is not needed yet -- compute it on another stream meanwhile."""
        assert img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3
        img = img.contiguous()
        self._enc_shape = (img.shape[0], img.shape[-1], img.device)
        self.handle.call('ch_sean_encode_features', img.data_ptr(), img.shape[0], img.shape[-1],
                         torch.cuda.current_stream(img.device).cuda_stream)

    def encode_regions(self, labels: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Second half: region-wise means of the feature map -> codes [B,19,512] (must be ordered after encode_features)."""
        B, S, dev = self._enc_shape
        assert labels.is_cuda and labels.dtype == torch.uint8 and tuple(labels.shape) == (B, S, S)
        labels = labels.contiguous()
        if out is None:
            out = torch.empty(B, 19, 512, dtype=torch.float32, device=dev)
        self.handle.call('ch_sean_encode_regions', labels.data_ptr(), out.data_ptr(), B, S,
                         torch.cuda.current_stream(dev).cuda_stream)
        return out

