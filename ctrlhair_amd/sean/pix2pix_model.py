"""Mirror of sean_codes/models/pix2pix_model.py::Pix2PixModel for the two inference modes CtrlHair uses
('UI_mode', 'style_code'; pix2pix_model.py:59-72) on top of the HIP library.

    sean_model(data, mode='UI_mode')   -> fake_image  cuda float32 [B,3,S,S]
    sean_model(data, mode='style_code') -> style codes cuda float32 [B,19,512]

`data` has the reference's keys: 'label' [B,1,S,S] (any numeric dtype), 'instance', 'image' [B,3,S,S], and 'obj_dic'
({str(j): {'ACE': tensor[512]}}, UI_mode) or 'path' (style_code, ignored).  Extension: data['noise'] (cuda float32
[B, noise_floats(S)]) pins the 18 noise planes the reference draws with torch.randn (normalization.py:111); without it
they are drawn on device from a seed taken from torch's global RNG (so torch.manual_seed makes runs repeatable).
"""
import torch

from .generator import SeanGenerator


class Pix2PixModel:
    def __init__(self, generator: SeanGenerator):
        self.netG = generator
        self.device = generator.device

    def eval(self):
        return self

    accepts_codes = True        # forward(mode='UI_mode') takes data['codes'] ([19,512]) in place of data['obj_dic'] (HairEditor.gen_img)

    def modules(self):          # change_status() walks .modules() looking for .status (hair_editor.py:33-36)
        return []

    def preprocess_input(self, data):
        """pix2pix_model.py:119-144 without the one-hot scatter: the library consumes uint8 labels directly."""
        label = data['label']
        if not isinstance(label, torch.Tensor):
            label = torch.as_tensor(label)
        label = label.to(self.device).long()
        assert label.dim() == 4 and label.shape[1] == 1, 'label must be [B,1,H,W]'
        return label[:, 0].to(torch.uint8).contiguous(), data.get('image')

    def forward(self, data, mode):
        labels, image = self.preprocess_input(data)
        B = labels.shape[0]
        if mode == 'UI_mode':
            if data.get('codes') is not None:           # extension: the [19,512] codes already assembled (HairEditor.gen_img)
                codes = torch.as_tensor(data['codes']).to(self.device).float().reshape(19, 512)
            else:
                obj_dic = data['obj_dic']
                codes = torch.stack([torch.as_tensor(obj_dic[str(j)]['ACE']).to(self.device).float().reshape(512)
                                     for j in range(19)])
            # the reference styles batch element 0 only (normalization.py:124); every sample gets that treatment here
            codes = codes[None].expand(B, 19, 512).contiguous()
            noise = data.get('noise')
            seed = 0 if noise is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            return self.netG.generate(labels, codes, noise, seed=seed)
        if mode == 'style_code':
            if not isinstance(image, torch.Tensor):
                image = torch.as_tensor(image)
            return self.netG.encode(image.to(self.device).float(), labels)
        raise ValueError("|mode| is invalid")

    __call__ = forward
