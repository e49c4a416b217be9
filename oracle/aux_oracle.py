"""ORACLE (test infrastructure, NOT product code) -- CPU restatements, in functional PyTorch fp32, of the three
smaller networks on the CtrlHair path: shape VAE, colour/texture MLPs, BiSeNet.  Same rules as sean_oracle.py:
only tests/, smoke() and bench.py's cpu_baseline may import it.  Pinned against golden vectors produced by the
imported reference modules (tests/golden/make_golden.py -> tests/golden/{shape,color,bisenet}_*.npz).
"""
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .sean_oracle import _t, to_torch  # noqa: F401

HAIR_IDX = 13   # global_value_utils.py:52


# ---- shape branch ------------------------------------------------------------------------------------------------

def pos_embedding(S: int = 256, order: int = 10) -> torch.Tensor:
    """shape_branch/model.py:18-30 generate_pos_embedding -> [4*order, S, S]."""
    c = np.linspace(0, 1, S, endpoint=False)
    bi = np.stack(np.meshgrid(c, c), 0)[None]
    nums = (2.0 ** np.arange(0, order) * np.pi)[:, None, None, None]
    g = np.concatenate([np.sin(nums * bi), np.cos(nums * bi)], axis=0).reshape(-1, S, S)
    return torch.tensor(g).float()


def my_layer_norm(x, gamma, beta, eps=1e-5):
    """my_torchlib/module.py:189-205: per-sample mean / unbiased std over C*H*W, (x-mean)/(std+eps)*gamma+beta."""
    B = x.shape[0]
    mean = x.reshape(B, -1).mean(1).reshape(B, 1, 1, 1)
    std = x.reshape(B, -1).std(1).reshape(B, 1, 1, 1)
    x = (x - mean) / (std + eps)
    return x * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)


def label_to_onehot19(labels):
    """shape_util.py:6-14 mask_label_to_one_hot (255 -> class 19, dropped)."""
    lab = _t(labels).long().unsqueeze(1).clone()
    lab[lab == 255] = 19
    B, _, H, W = lab.shape
    return torch.zeros(B, 20, H, W).scatter_(1, lab, 1.0)[:, :-1]


def split_hair_face(mask):
    """shape_util.py:23-26."""
    return mask[:, [HAIR_IDX]], torch.cat([mask[:, :HAIR_IDX], mask[:, HAIR_IDX + 1:]], dim=1)


def mask_encoder(sd, side, x):
    """MaskEncoder.forward (shape_branch/model.py:96-108) with Conv2dBlock(k4,s2,ZeroPad 1,'ln','lrelu')
    (my_torchlib/module.py:67-137); returns out_mean."""
    x = torch.cat([x, pos_embedding(x.shape[-1]).unsqueeze(0).expand(x.shape[0], -1, -1, -1)], dim=1)
    for l in range(7):
        p = f'{side}_encoder.layers.{l}'
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[p + '.conv.weight'], sd[p + '.conv.bias'], stride=2)
        x = F.leaky_relu(my_layer_norm(x, sd[p + '.norm.gamma'], sd[p + '.norm.beta']), 0.2)
    return F.linear(x.flatten(1), sd[f'{side}_encoder.out_layer.fc.weight'], sd[f'{side}_encoder.out_layer.fc.bias'])


def mask_decoder(sd, side, code):
    """MaskDecoder.forward (shape_branch/model.py:138-143)."""
    d = f'{side}_decoder'
    x = F.linear(code, sd[d + '.in_layer.fc.weight'], sd[d + '.in_layer.fc.bias']).reshape(-1, 2048, 2, 2)
    for l in range(7):
        p = f'{d}.layers.{2 * l + 1}'
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[p + '.conv.weight'], sd[p + '.conv.bias'])
        x = F.leaky_relu(my_layer_norm(x, sd[p + '.norm.gamma'], sd[p + '.norm.beta']), 0.2)
    return F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[d + '.out_layer.conv.weight'], sd[d + '.out_layer.conv.bias'])


@torch.no_grad()
def shape_encode(sd, labels):
    """ui/backend.py:81-86: labels uint8 [B,256,256] -> (hair_code [B,16], face_code [B,1024])."""
    hair, face = split_hair_face(label_to_onehot19(labels))
    return mask_encoder(sd, 'hair', hair), mask_encoder(sd, 'face', face)


@torch.no_grad()
def shape_decode(sd, hair_code, face_code):
    """forward_decode_by_code (shape_branch/model.py:175-199) + mask_one_hot_to_label (shape_util.py:17-20).
    Returns (hair_logit, face_logit, probs [B,19,H,W], labels uint8)."""
    hair_code, face_code = _t(hair_code), _t(face_code)
    hl = mask_decoder(sd, 'hair', torch.cat([face_code, hair_code], dim=1))
    fl = mask_decoder(sd, 'face', face_code)
    logit = torch.cat([fl[:, :HAIR_IDX], hl, fl[:, HAIR_IDX:]], dim=1)
    probs = torch.softmax(logit, dim=1)
    lab = torch.argmax(probs, dim=1)
    lab[probs.max(dim=1)[0] == 0] = 255
    return hl, fl, probs, lab.to(torch.uint8)


# ---- colour / texture branch -------------------------------------------------------------------------------------

@torch.no_grad()
def color_generate(gen, noise, cond):
    """EigenGenerator.forward (model_eigengan.py:62-84); cond = cat(noise_curliness, rgb_mean, pca_std)."""
    noise, cond = _t(noise), _t(cond)
    x = F.linear(cond, gen['main_layer_in.weight'], gen['main_layer_in.bias'])
    z = noise.reshape(noise.shape[0], 4, 2)
    for k in range(4):
        x = x + (gen[f'subspaces.{k}.L'] * z[:, k]) @ gen[f'subspaces.{k}.U'] + gen[f'subspaces.{k}.mu']
        x = F.linear(F.leaky_relu(x, 0.2), gen[f'main_layer_mid.{k}.1.weight'], gen[f'main_layer_mid.{k}.1.bias'])
    return x


@torch.no_grad()
def color_encode(dis, code):
    """Discriminator.net (model.py:94-111): 4 x LinearBlock(lrelu 0.2) + Linear -> [B,11]."""
    x = _t(code)
    for k in range(5):
        x = F.linear(x, dis[f'net.{k}.fc.weight'], dis[f'net.{k}.fc.bias'])
        if k < 4:
            x = F.leaky_relu(x, 0.2)
    return x


@torch.no_grad()
def color_predict(rgb, code):
    """Predictor.net (predictor_model.py:20-41): 3 x (Linear, BatchNorm1d eval, lrelu, dropout=id) + Linear -> [B,4]."""
    x = _t(code)
    for k in range(4):
        x = F.linear(x, rgb[f'net.{k}.fc.weight'], rgb[f'net.{k}.fc.bias'])
        if k < 3:
            p = f'net.{k}.norm'
            x = F.batch_norm(x, rgb[p + '.running_mean'], rgb[p + '.running_var'], rgb[p + '.weight'], rgb[p + '.bias'],
                             False, 0.1, 1e-5)
            x = F.leaky_relu(x, 0.2)
    return x


# ---- BiSeNet -----------------------------------------------------------------------------------------------------

BISENET_TO_CELEBA = [0, 1, 6, 7, 4, 5, 3, 8, 9, 15, 2, 10, 11, 12, 17, 16, 18, 13, 14]   # my_parsing_util.py:19-22,50-54


def _cbr(sd, p, x, stride=1, padding=1, bnname='bn', convname='conv'):
    """ConvBNReLU (model.py:14-35)."""
    x = F.conv2d(x, sd[f'{p}.{convname}.weight'], None, stride=stride, padding=padding)
    q = f'{p}.{bnname}'
    return F.relu(F.batch_norm(x, sd[q + '.running_mean'], sd[q + '.running_var'], sd[q + '.weight'], sd[q + '.bias'],
                               False, 0.1, 1e-5))


def _bn(sd, q, x):
    return F.batch_norm(x, sd[q + '.running_mean'], sd[q + '.running_var'], sd[q + '.weight'], sd[q + '.bias'],
                        False, 0.1, 1e-5)


def _basic_block(sd, p, x, stride):
    """BasicBlock.forward (resnet.py:36-48)."""
    r = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, stride=stride, padding=1)))
    r = _bn(sd, p + '.bn2', F.conv2d(r, sd[p + '.conv2.weight'], None, padding=1))
    sc = x
    if (p + '.downsample.0.weight') in sd:
        sc = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride=stride))
    return F.relu(sc + r)


def _arm(sd, p, x):
    """AttentionRefinementModule.forward (model.py:75-83)."""
    feat = _cbr(sd, p + '.conv', x)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(_bn(sd, p + '.bn_atten', F.conv2d(att, sd[p + '.conv_atten.weight'])))
    return feat * att


@torch.no_grad()
def bisenet_forward(sd, img, taps: Optional[dict] = None):
    """BiSeNet.forward output [0] (model.py:241-254) + argmax + swap_parsing_label_to_celeba_mask
    (my_parsing_util.py:45-54).  img f32 [B,3,H,W] normalised.  Returns (logits [B,19,H,W], labels uint8 CelebA ids)."""
    x = _t(img).float()
    H, W = x.shape[2:]
    # Resnet18.forward (resnet.py:71-80)
    x = F.relu(_bn(sd, 'cp.resnet.bn1', F.conv2d(x, sd['cp.resnet.conv1.weight'], None, stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for L in range(1, 5):
        for i in range(2):
            x = _basic_block(sd, f'cp.resnet.layer{L}.{i}', x, 2 if (i == 0 and L > 1) else 1)
        feats.append(x)
    _, feat8, feat16, feat32 = feats
    # ContextPath.forward (model.py:104-125)
    avg = _cbr(sd, 'cp.conv_avg', F.avg_pool2d(feat32, feat32.shape[2:]), padding=0)
    avg_up = F.interpolate(avg, feat32.shape[2:], mode='nearest')
    feat32_sum = _arm(sd, 'cp.arm32', feat32) + avg_up
    feat32_up = _cbr(sd, 'cp.conv_head32', F.interpolate(feat32_sum, feat16.shape[2:], mode='nearest'))
    feat16_sum = _arm(sd, 'cp.arm16', feat16) + feat32_up
    feat_cp8 = _cbr(sd, 'cp.conv_head16', F.interpolate(feat16_sum, feat8.shape[2:], mode='nearest'))
    if taps is not None:
        taps.update(feat8=feat8, feat16=feat16, feat32=feat32, feat_cp8=feat_cp8)
    # FeatureFusionModule.forward (model.py:198-210)
    feat = _cbr(sd, 'ffm.convblk', torch.cat([feat8, feat_cp8], dim=1), padding=0)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(F.conv2d(F.relu(F.conv2d(att, sd['ffm.conv1.weight'])), sd['ffm.conv2.weight']))
    fuse = feat * att + feat
    # BiSeNetOutput (model.py:43-46) + bilinear align_corners (model.py:250)
    out = F.conv2d(_cbr(sd, 'conv_out.conv', fuse), sd['conv_out.conv_out.weight'])
    logits = F.interpolate(out, (H, W), mode='bilinear', align_corners=True)
    parsing = logits.argmax(1)
    lut = torch.tensor(BISENET_TO_CELEBA, dtype=torch.uint8)
    return logits, lut[parsing]
