"""Test-only model provider: the same object surface as ctrlhair_amd.hair_editor.HipModels, but every network is the
CPU oracle.  Lets the host logic (HairEditor / Backend) run without a GPU ('Config 1: reference plumbing on CPU') and
gives the GPU tests an end-to-end A/B partner."""
import numpy as np
import torch

from ctrlhair_amd import models as M
from oracle import aux_oracle as A
from oracle import sean_oracle as O


class _Sean:
    def __init__(self, sd, ngf):
        self.sd, self.ngf, self.wc = O.to_torch(sd), ngf, {}
        self.device = torch.device('cpu')

    def eval(self):
        return self

    def modules(self):
        return []

    def __call__(self, data, mode):
        label = torch.as_tensor(data['label']).long()[:, 0].to(torch.uint8)
        B, S = label.shape[0], label.shape[-1]
        if mode == 'UI_mode':
            codes = torch.stack([torch.as_tensor(data['obj_dic'][str(j)]['ACE']).float().reshape(512) for j in range(19)])
            codes = codes[None].expand(B, 19, 512)
            from ctrlhair_amd.sean import arch
            noise = data.get('noise')
            if noise is None:
                noise = torch.randn(B, arch.noise_floats_per_sample(S, self.ngf))
            return O.generator_forward(self.sd, label, codes, torch.as_tensor(noise).cpu(), self.ngf, weights_cache=self.wc)
        if mode == 'style_code':
            return O.zencoder_forward(self.sd, torch.as_tensor(data['image']).float(), label)
        raise ValueError("|mode| is invalid")


class _Shape:
    def __init__(self, sd):
        self.sd = O.to_torch(sd)

    def encode_labels(self, labels):
        return A.shape_encode(self.sd, labels.reshape(-1, 256, 256).cpu())

    def decode_labels(self, hair_code, face_code):
        return A.shape_decode(self.sd, hair_code, face_code)[3]

    def forward_decode_by_code(self, hair_code, face_code):
        return A.shape_decode(self.sd, hair_code, face_code)[2]

    def forward_face_decoder(self, face_code):
        return A.mask_decoder(self.sd, 'face', face_code)

    def forward_decoder(self, hair_logit, face_logit):
        logit = torch.cat([face_logit[:, :13], hair_logit, face_logit[:, 13:]], dim=1)
        return torch.softmax(logit, dim=1)


class _Color:
    def __init__(self, cs):
        self.cs = {k: O.to_torch(v) for k, v in cs.items()}
        self.gen = lambda d: {'code': A.color_generate(self.cs['gen'], torch.as_tensor(d['noise']).float(), torch.cat(
            [torch.as_tensor(d['noise_curliness']).float(), torch.as_tensor(d['rgb_mean']).float(),
             torch.as_tensor(d['pca_std']).float()], 1))}
        self.dis = lambda d: (lambda o: {'adv': o[:, [0]], 'noise': o[:, 1:9], 'noise_curliness': o[:, 9:10]})(
            A.color_encode(self.cs['dis'], d['code']))
        self.rgb_model = lambda d: (lambda o: {'rgb_mean': o[:, :3], 'pca_std': o[:, 3:4]})(
            A.color_predict(self.cs['rgb'], d['code']))

    def edit_infer(self, hair_code, data):
        inner = self.dis({'code': hair_code})
        inner.update(data)
        return self.gen(inner)['code']


class _Parsing(M.FaceParsing):
    def __init__(self, sd):
        self.sd = O.to_torch(sd)
        self.device = torch.device('cpu')

    def parse_tensor(self, img, want_logits=False):
        lg, lab = A.bisenet_forward(self.sd, img.cpu())
        return lab, (lg if want_logits else None)


class OracleModels:
    def __init__(self, weights, ngf=64):
        self.device = torch.device('cpu')
        self.sean_model = _Sean(weights['sean'], ngf)
        self.mask_generator = _Shape(weights['shape'])
        self.solver_feature = _Color({'gen': weights['color_gen'], 'dis': weights['color_dis'], 'rgb': weights['color_rgb']})
        self.face_parsing = _Parsing(weights['bisenet'])
