"""CNN feature / context encoders (stay on PyTorch-ROCm / MIOpen per the north star).

Mirror of the reference's ``BasicEncoder`` parameter layout (``core/extractor.py:124-196``,
residual block ``:6-64``) so that reference checkpoints load key-for-key:
``conv1, norm1, layer{1,2,3}.{0,1}.{conv1,conv2,norm1,norm2[,norm3,downsample.{0,1}]}, conv2``.
Not part of the hand-written hot path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _make_norm(kind: str, ch: int) -> nn.Module:
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    if kind == "instance":
        return nn.InstanceNorm2d(ch)        # non-affine, no running stats -> contributes no keys
    if kind == "group":
        return nn.GroupNorm(ch // 8, ch)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"unknown norm_fn {kind!r}")


class ResidualBlock(nn.Module):
    def __init__(self, cin: int, cout: int, norm_fn: str, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.norm1 = _make_norm(norm_fn, cout)
        self.norm2 = _make_norm(norm_fn, cout)
        self.downsample = None
        if stride != 1:
            # the reference registers the same norm under two names (norm3 and downsample.1), so a
            # BatchNorm variant exposes both key sets (extractor.py:21-47)
            self.norm3 = _make_norm(norm_fn, cout)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class BasicEncoder(nn.Module):
    def __init__(self, output_dim: int = 128, norm_fn: str = "batch", dropout: float = 0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _make_norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, norm_fn, 1), ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, norm_fn, 2), ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, norm_fn, 2), ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = nn.Dropout2d(dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = torch.cat(list(x), dim=0)
        x = F.relu(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        if pair:
            return x[:n], x[n:]
        return x
