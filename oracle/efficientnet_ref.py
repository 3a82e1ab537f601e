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

    bf16_operand = False          # set by EfficientNetB0Ref(storage="bf16") on the stem and the 1x1 convs

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        pt, pb = same_pad(x.shape[2], k, s)
        pl, pr = same_pad(x.shape[3], k, s)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
        w = _wq(self.weight) if self.bf16_operand else self.weight
        return F.conv2d(x, w, self.bias, self.stride, 0, self.dilation, self.groups)


def swish(x):
    return x * torch.sigmoid(x)


# ---- bf16 activation storage (BASELINE configs[4]) as a property of the oracle ------------------------------------------
# The reference never runs reduced precision (utils/local_training.py:14 imports autocast and does not use it), so the bf16
# configuration has no reference counterpart.  What CAN be stated exactly is where that configuration rounds: every activation
# tensor the engine stores between kernels is bf16 (stem output, a0, y_e, a_e, y_d, a_s, y_p, block output, head conv output,
# head activation), so is the gradient stored at the same points on the way back, and the 1x1 / stem convolutions multiply
# bf16 copies of the fp32 master weights; everything else (accumulation, BatchNorm statistics, squeeze-excite vectors,
# depthwise weights, loss, Adam) is fp32.  storage="bf16" inserts exactly those roundings into this fp32 module, which makes
# "engine in bf16 mode vs oracle with the same storage points" a parity statement of its own (tests/test_golden_r4_gpu.py).
class _StoreBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _ident(x):
    return x


def _wq(w):
    """bf16 copy of an fp32 master weight as an MFMA operand; the gradient goes to the master (straight through)"""
    return w + (w.detach().to(torch.bfloat16).to(torch.float32) - w.detach())


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

    q = staticmethod(_ident)      # storage point: identity (fp32) or _StoreBF16.apply

    def forward(self, inputs, dc=None):
        q = self.q
        x = inputs
        if self.e != 1:
            x = q(swish(self._bn0(q(self._expand_conv(x)))))        # y_e, a_e
        x = swish(self._bn1(q(self._depthwise_conv(x))))            # y_d (a_d is never stored)
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(swish(self._se_reduce(sq)))
        x = q(torch.sigmoid(sq) * x)                                # a_s (stored, or rounded as the project conv's operand)
        x = self._bn2(q(self._project_conv(x)))                     # y_p
        if self.s == 1 and self.cin == self.cout:
            if dc is not None:
                x = x * dc.view(-1, 1, 1, 1)       # drop_connect: floor(keep + U) / keep per sample
            x = x + inputs
        return q(x)                                                 # block output


class EfficientNetB0Ref(nn.Module):
    """forward(x, dc=None, dropout=None) -> (feature[B,1280], logits[B,n_classes]).
    dc: [16, B] drop-connect multipliers (only the skip blocks use theirs); dropout: [B,1280]
    multipliers (0 or 1/(1-p)) applied to the pooled feature before `_fc`."""

    def __init__(self, n_classes, storage="fp32"):
        super().__init__()
        assert storage in ("fp32", "bf16")
        self.storage = storage
        self._conv_stem = SameConv(3, 32, 3, 2, bias=False)
        self._bn0 = nn.BatchNorm2d(32, momentum=BN_MOM, eps=BN_EPS)
        self._blocks = nn.ModuleList([MBConv(*a) for a in block_args()])
        self._conv_head = SameConv(320, 1280, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(1280, momentum=BN_MOM, eps=BN_EPS)
        self._fc = nn.Linear(1280, n_classes)
        if storage == "bf16":
            for m in self.modules():
                if isinstance(m, SameConv) and m.groups == 1 and m.bias is None:      # stem, expand, project, head convs
                    m.bf16_operand = True
                if isinstance(m, MBConv):
                    m.q = _StoreBF16.apply

    def forward(self, x, dc=None, dropout=None):
        q = _StoreBF16.apply if self.storage == "bf16" else _ident
        if self.storage == "bf16":
            x = x.to(torch.bfloat16).to(torch.float32)              # the stem reads a bf16 im2col of the input batch
        x = q(swish(self._bn0(q(self._conv_stem(x)))))
        for i, blk in enumerate(self._blocks):
            x = blk(x, None if dc is None else dc[i])
        x = q(swish(self._bn1(q(self._conv_head(x)))))
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
