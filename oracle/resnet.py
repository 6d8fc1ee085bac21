"""Oracle restatement of the torchvision ResNet-50/101 body (TEST INFRASTRUCTURE).

The reference builds its backbone with ``torchvision.models.<name>(replace_stride_with_dilation=
[False, False, dilation], norm_layer=FrozenBatchNorm2d)`` (models/backbone.py:94-96).  torchvision is
a third-party dependency that is absent from the reference tree and not installed in this image
(version unpinned by the reference): **parity unpinned** -- this file restates the public
architecture (Bottleneck v1.5: stride on the 3x3 conv; 7x7/2 stem + 3x3/2 max-pool; stage widths
64/128/256/512 x expansion 4; blocks [3,4,6,3] / [3,4,23,3]) with torchvision's state_dict names
(``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``).
It is also what ``tests/golden/make_fixtures.py`` plugs into the reference's ``Backbone`` class in
place of the missing torchvision when it generates the end-to-end golden vectors.
"""
import torch
from torch import nn

_BLOCKS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, name, replace_stride_with_dilation=(False, False, False), norm_layer=nn.BatchNorm2d):
        super().__init__()
        blocks = _BLOCKS[name]
        self.inplanes, self.dilation = 64, 1
        self._norm = norm_layer
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, blocks[0], 1, False)
        self.layer2 = self._stage(128, blocks[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._stage(256, blocks[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._stage(512, blocks[3], 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, n, stride, dilate):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 self._norm(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, prev_dil, down, self._norm)]
        self.inplanes = planes * 4
        for _ in range(1, n):
            layers.append(Bottleneck(self.inplanes, planes, 1, self.dilation, None, self._norm))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(**kw):
    kw.pop("pretrained", None)
    return ResNet("resnet50", **kw)


def resnet101(**kw):
    kw.pop("pretrained", None)
    return ResNet("resnet101", **kw)


class IntermediateLayerGetter(nn.ModuleDict):
    """Same contract as torchvision.models._utils.IntermediateLayerGetter: run the children in
    registration order, collect the outputs named in ``return_layers``, drop children after the last."""

    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        remaining = dict(return_layers)
        layers = {}
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = {}
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
