"""Static description of the SEAN generator the hot path implements.

This is the architecture table only (no arithmetic): block list, ACE execution
order, channel widths and the reference state-dict key families.  It mirrors

  * sean_codes/models/networks/generator.py:24-54   (block list, ngf multipliers)
  * sean_codes/models/networks/architecture.py:21-96 (ResBlock: ace_s/ace_0/ace_1, conv_s/conv_0/conv_1)
  * sean_codes/models/networks/normalization.py:71-106 (per-ACE parameters)

and is shared by the procedural weight generator, the weight loader that packs
tensors for the HIP library, the oracle and the tests.
"""
from dataclasses import dataclass
from typing import List

LABEL_NC = 19          # sean_codes/options/test_options.py + global_value_utils.py:49-52
STYLE_LEN = 512        # normalization.py:79
SPADE_HIDDEN = 128     # normalization.py:237
NUM_UP = 5             # generator.py:57-58 ('normal')


@dataclass(frozen=True)
class AceSpec:
    name: str          # e.g. 'up_0.ace_s' (state-dict prefix)
    channels: int      # norm_nc
    styled: bool       # use_rgb (generator.py:43: up_3 is SPADE-only)
    res_div: int       # ACE runs at S // res_div
    index: int         # execution order 0..17 (noise plane index)


@dataclass(frozen=True)
class BlockSpec:
    name: str
    fin: int
    fout: int
    res_div: int       # block runs at S // res_div
    up_before: bool    # nn.Upsample(x2) applied to the block input (generator.py:85-100)
    styled: bool

    @property
    def fmid(self) -> int:
        return min(self.fin, self.fout)

    @property
    def learned_shortcut(self) -> bool:
        return self.fin != self.fout


def blocks(ngf: int = 64) -> List[BlockSpec]:
    nf = ngf
    return [
        BlockSpec('head_0', 16 * nf, 16 * nf, 32, False, True),
        BlockSpec('G_middle_0', 16 * nf, 16 * nf, 16, True, True),
        BlockSpec('G_middle_1', 16 * nf, 16 * nf, 16, False, True),
        BlockSpec('up_0', 16 * nf, 8 * nf, 8, True, True),
        BlockSpec('up_1', 8 * nf, 4 * nf, 4, True, True),
        BlockSpec('up_2', 4 * nf, 2 * nf, 2, True, True),
        BlockSpec('up_3', 2 * nf, 1 * nf, 1, True, False),
    ]


def aces(ngf: int = 64) -> List[AceSpec]:
    """ACE instances in *execution* order: shortcut first (architecture.py:71), then ace_0, ace_1."""
    out: List[AceSpec] = []
    for blk in blocks(ngf):
        if blk.learned_shortcut:
            out.append(AceSpec(blk.name + '.ace_s', blk.fin, blk.styled, blk.res_div, len(out)))
        out.append(AceSpec(blk.name + '.ace_0', blk.fin, blk.styled, blk.res_div, len(out)))
        out.append(AceSpec(blk.name + '.ace_1', blk.fmid, blk.styled, blk.res_div, len(out)))
    return out


def noise_plane_sizes(S: int, ngf: int = 64) -> List[int]:
    """Side length of each ACE's noise plane (normalization.py:111 draws randn(B, W, H, 1))."""
    return [S // a.res_div for a in aces(ngf)]


def noise_floats_per_sample(S: int, ngf: int = 64) -> int:
    return sum(r * r for r in noise_plane_sizes(S, ngf))
