"""detectron2 ResNet-50 backbone for the R50 configs ("use torch/MIOpen convs, no custom kernels").

Mirror of detectron2/modeling/backbone/resnet.py (BasicStem :330-359, BottleneckBlock :100-212, ResNet :362-470) and
detectron2/layers/batch_norm.py FrozenBatchNorm2d (:13-65) with the reference's parameter/buffer names
(stem.conv1.{weight,norm.*}, res{2..5}.{i}.{conv1,conv2,conv3,shortcut}.{weight,norm.*}); STRIDE_IN_1X1 False, FrozenBN,
outputs res3/res4/res5.  Convolutions are library (MIOpen) calls; in the 16-bit policies they run channels-last.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n) - eps)

    def forward(self, x):
        # folded affine in fp32, applied in the activation dtype (== F.batch_norm(training=False))
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + shift.to(x.dtype).view(1, -1, 1, 1)


class NormConv(nn.Conv2d):
    """detectron2.layers.Conv2d(bias=False, norm=FrozenBN)."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.norm = FrozenBatchNorm2d(cout)

    folded = None      # (weight * bn_scale, bn_shift) in the policy dtype, built by ResNet50.cast_weights()

    def forward(self, x):
        if self.folded is not None:            # frozen BN is an affine map per output channel: one conv + bias pass
            return self._conv_forward(x, self.folded[0], self.folded[1])
        return self.norm(self._conv_forward(x, self.weight, None))

    def fold(self, dtype):
        n = self.norm
        scale = n.weight.float() * (n.running_var.float() + n.eps).rsqrt()
        shift = n.bias.float() - n.running_mean.float() * scale
        w = (self.weight.float() * scale.view(-1, 1, 1, 1)).to(dtype)
        if dtype != torch.float32:
            w = w.contiguous(memory_format=torch.channels_last)
        self.folded = (w, shift.to(dtype))


class BasicStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = NormConv(3, 64, 7, stride=2, padding=3)

    def forward(self, x):
        return F.max_pool2d(F.relu(self.conv1(x)), kernel_size=3, stride=2, padding=1)


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, mid, stride):
        super().__init__()
        self.shortcut = NormConv(cin, cout, 1, stride=stride) if cin != cout else None
        self.conv1 = NormConv(cin, mid, 1)
        self.conv2 = NormConv(mid, mid, 3, stride=stride, padding=1)
        self.conv3 = NormConv(mid, cout, 1)

    def forward(self, x):
        y = F.relu(self.conv1(x))
        y = F.relu(self.conv2(y))
        y = self.conv3(y)
        return F.relu(y + (x if self.shortcut is None else self.shortcut(x)))


class ResNet50(nn.Module):
    size_divisibility = 32

    def __init__(self, precision):
        super().__init__()
        self.stem = BasicStem()
        cin = 64
        for name, n, cout, stride in (("res2", 3, 256, 1), ("res3", 4, 512, 2), ("res4", 6, 1024, 2), ("res5", 3, 2048, 2)):
            blocks = []
            for b in range(n):
                blocks.append(BottleneckBlock(cin, cout, cout // 4, stride if b == 0 else 1))
                cin = cout
            setattr(self, name, nn.Sequential(*blocks))
        self.precision = precision
        self._out_feature_channels = {"res3": 512, "res4": 1024, "res5": 2048}
        self._out_feature_strides = {"res3": 8, "res4": 16, "res5": 32}

    def forward(self, x):
        gd = self.precision.gemm
        x = x.to(gd)
        if gd != torch.float32:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.res2(self.stem(x))
        out = {}
        for name in ("res3", "res4", "res5"):
            x = getattr(self, name)(x)
            out[name] = x.float()
        return out

    def cast_weights(self):
        """after the weights are loaded: fold every FrozenBatchNorm2d into its convolution (weights are frozen at inference)
        and put the folded weights in the policy dtype (channels-last for the 16-bit policies)."""
        gd = self.precision.gemm
        for m in self.modules():
            if isinstance(m, NormConv):
                m.fold(gd)
        return self

    def output_shape(self):
        return {k: dict(channels=self._out_feature_channels[k], stride=self._out_feature_strides[k]) for k in ("res3", "res4", "res5")}
