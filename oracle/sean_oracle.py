"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the SEAN generator
forward of XuyangGuo/CtrlHair, in plain functional PyTorch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (ctrlhair_amd + libctrlhair_hip.so) never does and fails loudly without its
HIP library.

Why torch and not numpy/C: every arithmetic op on this path in the reference *is* an ATen CPU
kernel of the torch build installed in this image (torch 2.10: conv2d, batch_norm, linear,
interpolate, tanh -- SURVEY.md 8c "third-party arithmetic"); calling the same functional ops
restates the reference's algorithm with the reference's own rounding behaviour.  The restatement
is deliberately the *dense* algorithm of the reference (one-hot conv, broadcast style map, dense
3x3 style convs), i.e. it does not share the label-LUT reformulation the HIP path uses.

Pinning: the reference has no tests/golden vectors for this path, so this oracle is
pinned against outputs of the imported reference modules themselves:
tests/golden/make_golden.py (run in the build container where /root/reference exists) ->
tests/golden/sean_*.npz, checked by tests/test_oracle_golden.py, plus a direct module-vs-oracle
comparison in tests/test_oracle_vs_reference.py (skipped where /root/reference is absent).

Each function cites the reference lines it follows.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

LABEL_NC = 19
STYLE_LEN = 512


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def to_torch(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: _t(v) for k, v in sd.items()}


def spectral_weight(sd, prefix):
    """torch.nn.utils.spectral_norm in eval mode: W = W_orig / (u . (W_mat v)), no power
    iteration (torch/nn/utils/spectral_norm.py compute_weight(do_power_iteration=False));
    applied to conv_0/conv_1/conv_s by architecture.py:42-46."""
    w = sd[prefix + '.weight_orig']
    u, v = sd[prefix + '.weight_u'], sd[prefix + '.weight_v']
    sigma = torch.dot(u, torch.mv(w.reshape(w.shape[0], -1), v))
    return w / sigma


def one_hot(labels: torch.Tensor) -> torch.Tensor:
    """pix2pix_model.py:133-138: scatter_ of the label map into a [B,19,H,W] fp32 one-hot."""
    lab = labels.long().unsqueeze(1)
    B, _, H, W = lab.shape
    # ids >= 19 ("no class", e.g. 255) make scatter_ raise in the reference; the library defines them as an all-zero
    # one-hot (include/ctrlhair_hip.h), restated here through a 20th channel that is dropped
    oh = torch.zeros(B, LABEL_NC + 1, H, W, dtype=torch.float32).scatter_(1, lab.clamp(max=LABEL_NC), 1.0)
    return oh[:, :LABEL_NC].contiguous()


def spade(sd, p, segmap):
    """normalization.py:249-257 (SPADE.forward): relu(conv 19->128) then two 128->C convs."""
    actv = F.relu(F.conv2d(segmap, sd[p + '.mlp_shared.0.weight'], sd[p + '.mlp_shared.0.bias'], padding=1))
    gamma = F.conv2d(actv, sd[p + '.mlp_gamma.weight'], sd[p + '.mlp_gamma.bias'], padding=1)
    beta = F.conv2d(actv, sd[p + '.mlp_beta.weight'], sd[p + '.mlp_beta.bias'], padding=1)
    return gamma, beta


def ace(sd, p, x, seg, codes, noise_plane, styled, stats_out=None):
    """normalization.py:108-189 (ACE.forward), batched 'else' branch semantics == UI_mode arithmetic
    applied to every sample (SURVEY.md 7 'UI_mode is batch-1 only').

    noise_plane: [B, W, H] -- the tensor randn(B, W, H, 1) of normalization.py:111 without its
    trailing 1; ``(n * noise_var).transpose(1, 3)`` gives added[b,c,h,w] = n[b,w,h] * noise_var[c].
    """
    B, C, H, W = x.shape
    added = (noise_plane.reshape(B, W, H, 1) * sd[p + '.noise_var']).transpose(1, 3)
    xin = x + added
    if stats_out is not None:  # calibration mode (procedural weights only): record & use batch stats
        mean = xin.mean(dim=(0, 2, 3))
        var = xin.var(dim=(0, 2, 3), unbiased=False)
        stats_out[p + '.param_free_norm.running_mean'] = mean.numpy().copy()
        stats_out[p + '.param_free_norm.running_var'] = var.numpy().copy()
        sd[p + '.param_free_norm.running_mean'] = mean
        sd[p + '.param_free_norm.running_var'] = var
    # sync_batchnorm/batchnorm.py:52-55, eval: F.batch_norm with running stats, affine=False, eps 1e-5
    normalized = F.batch_norm(xin, sd[p + '.param_free_norm.running_mean'], sd[p + '.param_free_norm.running_var'],
                              None, None, False, 0.1, 1e-5)
    segmap = F.interpolate(seg, size=(H, W), mode='nearest')          # normalization.py:115
    gamma_spade, beta_spade = spade(sd, p + '.Spade', segmap)         # :175
    if not styled:                                                    # :183-187 (use_rgb False)
        return normalized * (1 + gamma_spade) + beta_spade
    # :117-153 -- middle_avg[b,:,p] = relu(fc_mu_j(code[b,j])) for j = label(p) (only labels with >=1 px
    # at this resolution are ever written; all others stay zero -- and are never read back).
    lab = segmap.argmax(dim=1)                                        # [B,H,W]
    mu = torch.zeros(B, LABEL_NC, STYLE_LEN)
    for j in range(LABEL_NC):
        mu[:, j] = F.relu(F.linear(codes[:, j], sd[f'{p}.fc_mu{j}.weight'], sd[f'{p}.fc_mu{j}.bias']))
    middle_avg = torch.gather(mu, 1, lab.reshape(B, H * W, 1).expand(B, H * W, STYLE_LEN))
    middle_avg = middle_avg.reshape(B, H, W, STYLE_LEN).permute(0, 3, 1, 2).contiguous()
    middle_avg = middle_avg * segmap.sum(dim=1, keepdim=True)         # pixels of no class are never written (:127-129)
    gamma_avg = F.conv2d(middle_avg, sd[p + '.conv_gamma.weight'], sd[p + '.conv_gamma.bias'], padding=1)  # :172
    beta_avg = F.conv2d(middle_avg, sd[p + '.conv_beta.weight'], sd[p + '.conv_beta.bias'], padding=1)    # :173
    ga = torch.sigmoid(sd[p + '.blending_gamma'])                     # :177-178
    ba = torch.sigmoid(sd[p + '.blending_beta'])
    gamma_final = ga * gamma_avg + (1 - ga) * gamma_spade             # :180-181
    beta_final = ba * beta_avg + (1 - ba) * beta_spade
    return normalized * (1 + gamma_final) + beta_final                # :182


def resblock(sd, blk, x, seg, codes, noise_iter, weights_cache, stats_out=None, taps=None):
    """architecture.py:69-96 (SPADEResnetBlock.forward / shortcut / actvn)."""
    name, styled = blk.name, blk.styled

    def w(prefix):
        if prefix not in weights_cache:
            weights_cache[prefix] = spectral_weight(sd, prefix)
        return weights_cache[prefix]

    def tap(k, v):
        if taps is not None:
            taps[name + k] = v

    if blk.learned_shortcut:
        x_s = ace(sd, name + '.ace_s', x, seg, codes, next(noise_iter), styled, stats_out)
        tap('.hs', x_s)
        x_s = F.conv2d(x_s, w(name + '.conv_s'))
        tap('.xs', x_s)
    else:
        x_s = x
    dx = F.leaky_relu(ace(sd, name + '.ace_0', x, seg, codes, next(noise_iter), styled, stats_out), 0.2)
    tap('.h0', dx)
    dx = F.conv2d(dx, w(name + '.conv_0'), sd[name + '.conv_0.bias'], padding=1)
    tap('.dx', dx)
    dx = F.leaky_relu(ace(sd, name + '.ace_1', dx, seg, codes, next(noise_iter), styled, stats_out), 0.2)
    tap('.h1', dx)
    dx = F.conv2d(dx, w(name + '.conv_1'), sd[name + '.conv_1.bias'], padding=1)
    return x_s + dx


@torch.no_grad()
def zencoder_forward(sd: Dict[str, torch.Tensor], img, labels, taps: Optional[dict] = None) -> torch.Tensor:
    """architecture.py:155-207 (Zencoder.__init__ layer list + forward), fed like
    Pix2PixModel.forward(mode='style_code') (pix2pix_model.py:69-72).
    img f32 [B,3,S,S]; labels uint8 [B,S,S] -> codes [B,19,512]."""
    img, labels = _t(img).float(), _t(labels)
    p = 'Zencoder.model.'
    x = F.conv2d(F.pad(img, (1, 1, 1, 1), mode='reflect'), sd[p + '1.weight'], sd[p + '1.bias'])      # :158-159
    x = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2)
    for i in (4, 7):                                                                                  # :161-164
        x = F.conv2d(x, sd[f'{p}{i}.weight'], sd[f'{p}{i}.bias'], stride=2, padding=1)
        x = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2)
    x = F.conv_transpose2d(x, sd[p + '10.weight'], sd[p + '10.bias'], stride=2, padding=1, output_padding=1)  # :169-172
    x = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2)
    codes = torch.tanh(F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), sd[p + '14.weight'], sd[p + '14.bias']))  # :174
    if taps is not None:
        taps['zenc.feat'] = codes
    seg = F.interpolate(one_hot(labels), size=codes.shape[2:], mode='nearest')                        # :181
    B, Fd = codes.shape[:2]
    out = torch.zeros(B, LABEL_NC, Fd)
    for b in range(B):                                                                                # :195-203
        for j in range(LABEL_NC):
            m = seg[b, j].bool()
            n = int(m.sum())
            if n > 0:
                out[b, j] = codes[b].masked_select(m).reshape(Fd, n).mean(1)
    return out


def split_noise(noise: torch.Tensor, S: int, ngf: int) -> List[torch.Tensor]:
    """[B, NF] flat noise (ctrlhair_amd.procedural.noise_planes layout) -> 18 planes [B, W, H]."""
    from ctrlhair_amd.sean import arch
    planes, off = [], 0
    for r in arch.noise_plane_sizes(S, ngf):
        planes.append(noise[:, off:off + r * r].reshape(-1, r, r))
        off += r * r
    assert off == noise.shape[1]
    return planes


@torch.no_grad()
def generator_forward(sd: Dict[str, torch.Tensor], labels, codes, noise, ngf: int = 64,
                      stats_out: Optional[dict] = None, taps: Optional[dict] = None,
                      weights_cache: Optional[dict] = None) -> torch.Tensor:
    """generator.py:72-109 (SPADEGenerator.forward) fed by pix2pix_model.py:119-144 (one-hot).

    labels uint8 [B,S,S]; codes f32 [B,19,512]; noise f32 [B,NF] -> image f32 [B,3,S,S] in [-1,1].
    """
    from ctrlhair_amd.sean import arch
    labels, codes, noise = _t(labels), _t(codes).float(), _t(noise).float()
    B, S = labels.shape[0], labels.shape[-1]
    seg = one_hot(labels)
    planes = iter(split_noise(noise, S, ngf))
    wc = {} if weights_cache is None else weights_cache
    sw = S // (2 ** arch.NUM_UP)                                      # generator.py:56-70
    x = F.interpolate(seg, size=(sw, sw))                             # :75 (nearest)
    x = F.conv2d(x, sd['fc.weight'], sd['fc.bias'], padding=1)        # :76
    if taps is not None:
        taps['fc'] = x
    for blk in arch.blocks(ngf):
        if blk.up_before:
            x = F.interpolate(x, scale_factor=2, mode='nearest')      # nn.Upsample(scale_factor=2), :53
        x = resblock(sd, blk, x, seg, codes, planes, wc, stats_out, taps)
        if taps is not None:
            taps[blk.name] = x
    x = F.conv2d(F.leaky_relu(x, 0.2), sd['conv_img.weight'], sd['conv_img.bias'], padding=1)  # :107
    if stats_out is not None:
        gain = 0.5 / float(x.std())
        stats_out['conv_img.gain'] = np.float32(gain)
        x = x * gain   # bias scaled too here; make_calibration re-runs a clean forward afterwards
    return torch.tanh(x)                                              # :108
