"""Oracle EfficientNet-B0: the efficientnet-pytorch 0.7.1 topology restated with plain torch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What it restates: the model the reference builds for `--model Efficient_b0`
(model/all_models.py:73-75 -> model/efficientnet.py:28-33 ->
efficientnet_pytorch.EfficientNet.from_pretrained('efficientnet-b0')) with `_fc` swapped for
nn.Linear(1280, n_classes) (model/all_models.py:121-124), in the patched form the trainer needs:
forward(x) -> (feature[B,1280], logits[B,C]).  efficientnet-pytorch==0.7.1
(requirements.txt:15) is NOT vendored in the reference and not installed here, so this is
"parity unpinned": the restatement follows the package's published structure:
static TF-"same" padding, BN eps 1e-3 / momentum 0.01, Swish = x*sigmoid(x), squeeze-excite with
max(1, int(0.25*block_input)) channels, drop-connect p = 0.2*idx/16 on the residual branch,
dropout 0.2 before the classifier.  Parameter names/order follow the package's state_dict.

Training-time randomness is made explicit: forward() takes the per-sample drop-connect
multipliers and the dropout multiplier mask as tensors (None = identity), so the HIP engine and
the oracle can be driven with the same draws.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

# (repeats, kernel, stride, expand, in, out)
B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
             (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
BN_EPS, BN_MOM = 1e-3, 0.01
DROP_CONNECT, DROPOUT = 0.2, 0.2


def block_args():
    """[(kernel, stride, expand, cin, cout)] for the 16 MBConv blocks."""
    out = []
    for r, k, s, e, i, o in B0_STAGES:
        out.append((k, s, e, i, o))
        for _ in range(r - 1):
            out.append((k, 1, e, o, o))
    return out


def same_pad(size, k, s):
    """static same padding (left/top, right/bottom) for an input of `size`."""
    o = math.ceil(size / s)
    p = max((o - 1) * s + k - size, 0)
    return p // 2, p - p // 2


class SameConv(nn.Conv2d):
    """Conv2dStaticSamePadding with the padding computed from the actual input size."""

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        pt, pb = same_pad(x.shape[2], k, s)
        pl, pr = same_pad(x.shape[3], k, s)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


def swish(x):
    return x * torch.sigmoid(x)


class MBConv(nn.Module):
    def __init__(self, k, s, e, cin, cout):
        super().__init__()
        self.k, self.s, self.e, self.cin, self.cout = k, s, e, cin, cout
        ce = cin * e
        if e != 1:
            self._expand_conv = SameConv(cin, ce, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(ce, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = SameConv(ce, ce, k, s, groups=ce, bias=False)
        self._bn1 = nn.BatchNorm2d(ce, momentum=BN_MOM, eps=BN_EPS)
        cs = max(1, int(cin * 0.25))
        self._se_reduce = SameConv(ce, cs, 1)
        self._se_expand = SameConv(cs, ce, 1)
        self._project_conv = SameConv(ce, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, momentum=BN_MOM, eps=BN_EPS)

    def forward(self, inputs, dc=None):
        x = inputs
        if self.e != 1:
            x = swish(self._bn0(self._expand_conv(x)))
        x = swish(self._bn1(self._depthwise_conv(x)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(swish(self._se_reduce(sq)))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        if self.s == 1 and self.cin == self.cout:
            if dc is not None:
                x = x * dc.view(-1, 1, 1, 1)       # drop_connect: floor(keep + U) / keep per sample
            x = x + inputs
        return x


class EfficientNetB0Ref(nn.Module):
    """forward(x, dc=None, dropout=None) -> (feature[B,1280], logits[B,n_classes]).
    dc: [16, B] drop-connect multipliers (only the skip blocks use theirs); dropout: [B,1280]
    multipliers (0 or 1/(1-p)) applied to the pooled feature before `_fc`."""

    def __init__(self, n_classes):
        super().__init__()
        self._conv_stem = SameConv(3, 32, 3, 2, bias=False)
        self._bn0 = nn.BatchNorm2d(32, momentum=BN_MOM, eps=BN_EPS)
        self._blocks = nn.ModuleList([MBConv(*a) for a in block_args()])
        self._conv_head = SameConv(320, 1280, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(1280, momentum=BN_MOM, eps=BN_EPS)
        self._fc = nn.Linear(1280, n_classes)

    def forward(self, x, dc=None, dropout=None):
        x = swish(self._bn0(self._conv_stem(x)))
        for i, blk in enumerate(self._blocks):
            x = blk(x, None if dc is None else dc[i])
        x = swish(self._bn1(self._conv_head(x)))
        feature = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        h = feature if dropout is None else feature * dropout
        return feature, self._fc(h)


def draw_stochastic(B, generator=None):
    """One training-step draw of (dc[16,B], dropout[B,1280]) with efficientnet-pytorch's formulas:
    block idx uses p = 0.2*idx/16, mask = floor(1-p + U[0,1)), out = x/(1-p)*mask; dropout p = 0.2."""
    dc = torch.ones((16, B))
    for idx in range(16):
        p = DROP_CONNECT * idx / 16.0
        if p > 0:
            keep = 1.0 - p
            dc[idx] = torch.floor(keep + torch.rand(B, generator=generator)) / keep
    dr = (torch.rand((B, 1280), generator=generator) >= DROPOUT).float() / (1.0 - DROPOUT)
    return dc, dr
