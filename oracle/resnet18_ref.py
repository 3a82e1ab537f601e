"""Oracle ResNet-18: torchvision-0.13 topology restated with plain torch.nn.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What it restates: the model object the reference builds at
``model/build_model.py:5-10`` -> ``model/all_models.py:53-54`` (torchvision
``resnet18``) with the last layer swapped for ``nn.Linear(512, n_classes)``
(``model/all_models.py:117-120``), in the locally patched form the trainer
expects: ``forward(x) -> (feature[B,512], logits[B,C])``
(``utils/local_training.py:657, 937, 983, 1030, 1178``).  torchvision is not
vendored in the reference and not installed here, so this is "parity
unpinned" at the model boundary; parameter names and order follow the
torchvision state_dict (conv1, bn1, layer{1..4}.{0,1}.*, fc).
"""
import torch
from torch import nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, 0, bias=False),
                nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + idt)


class ResNet18Ref(nn.Module):
    """forward(x[B,3,H,W]) -> (feature[B,512], logits[B,n_classes])."""

    WIDTHS = (64, 128, 256, 512)

    def __init__(self, n_classes):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, w in enumerate(self.WIDTHS, start=1):
            stride = 1 if li == 1 else 2
            layer = nn.Sequential(BasicBlock(cin, w, stride), BasicBlock(w, w, 1))
            setattr(self, f"layer{li}", layer)
            cin = w
        self.fc = nn.Linear(512, n_classes)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        feature = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return feature, self.fc(feature)
