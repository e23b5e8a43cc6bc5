"""ResNet-50/101 backbone of UPSNet for the MI355X inference path.

What it computes is fixed by the reference (upsnet/models/resnet.py:53-175, 314-356): a 7x7/2 stem +
3x3/2 max-pool, then four stages of caffe-style bottlenecks (the stride sits on the FIRST 1x1), every
BatchNorm frozen, optionally a deformable 3x3 (DCN v1, zero-initialised 18-channel offset conv) in
the res3..res5 blocks. Parameter names are the reference's (`resnet_backbone.res{2..5}.layers.N.conv{1,2,3}`,
`bn{1,2,3}`, `conv2_offset`, `downsample.{0,1}`, `conv1.conv1/bn1`) so its checkpoints load unchanged.

How it is built here is table-driven: one block class covers both bottleneck kinds, and because
every BN is frozen the whole backbone can be re-parameterised for inference (`fold_frozen_bn`): the BN
affine is folded into the preceding convolution, removing one read+write of every activation.
Every convolution runs on the hand-written MFMA kernels (models/hipconv.py); the stem convolution, its ReLU and the 3x3/2 max-pool are
one launch (csrc/stem_pool.hip, csrc/stem_pool_bf16.hip).
"""
import warnings

import torch
import torch.nn as nn

from ..config.config import config
from ..operators.modules.deform_conv import DeformConv
from . import hipconv


def _frozen_bn(ch):
    bn = nn.BatchNorm2d(ch)
    bn.eval()
    for p in bn.parameters():
        p.requires_grad = False
    return bn


class _Block(nn.Module):
    """1x1(stride) -> 3x3 (plain or deformable) -> 1x1(x4), each followed by frozen BN; residual add; ReLU."""
    expansion = 4
    deformable = False
    # (r13) True on a plain bottleneck that lies UPSTREAM of a deformable one (ResNetBackbone sets it): its 3x3 stays on the F(2x2) Winograd
    # form. The F(4x4) form's rounding error is 3-4x larger, and a chain of deformable layers multiplies an upstream difference by the
    # local feature gradient per layer -- UPSNet-101-DCN at 1024x2048: the 30th offset prediction at 1.069 of the 1e-4 bound with res2's
    # three 3x3 layers on F(4x4), 0.918 with them on F(2x2) (profiles/r12_parity.txt, r11_parity.txt). Costs 3 x ~13 us there, nothing on
    # UPSNet-50 (no deformable bottleneck in its backbone).
    feeds_deformable = False

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, fix_bn=True, deformable_group=1):
        super().__init__()
        make_bn = _frozen_bn if fix_bn else nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = make_bn(planes)
        if self.deformable:
            self.conv2_offset = nn.Conv2d(planes, 18 * deformable_group, 3, padding=1)
            nn.init.zeros_(self.conv2_offset.weight)
            nn.init.zeros_(self.conv2_offset.bias)
            self.conv2 = DeformConv(planes, planes, 3, stride=1, padding=dilation, dilation=dilation, bias=False)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = make_bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = make_bn(planes * self.expansion)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if isinstance(self.bn1, nn.Identity):  # BN folded: conv + bias (+ residual) + ReLU in one kernel each
            return self.forward_chain(x)[0]
        y = torch.relu_(self.bn1(self.conv1(x)))
        y = self.conv2(y, self.conv2_offset(y)) if self.deformable else self.conv2(y)
        y = torch.relu_(self.bn2(y))
        y = self.bn3(self.conv3(y))
        y += x if self.downsample is None else self.downsample(x)
        return torch.relu_(y)


    def forward_chain(self, x, y1=None, nxt=None):
        """Folded-BN forward as a link of a stage: y1 = this block's conv1 output if the previous block already produced it,
        nxt = the next block of the stage. Returns (block output, the next block's conv1 output or None): where the two 1x1 layers
        at a block boundary are HBM-bound (res2), conv3 + shortcut + ReLU of this block and conv1 + ReLU of the next run as one
        launch that never reads the block output back (hipconv.use_pair, csrc/conv1x1_pair.hip; bit-identical results)."""
        # bf16 mode: activations stay bf16 between the layers of the backbone (hipconv.act_dtype); a deformable 3x3 and its offset
        # predictor read fp32, so the 1x1 in front of them writes fp32 there
        if y1 is None and hipconv.use_block(self, x):      # bf16 mode: the whole identity block as one launch
            return hipconv.block(self, x), None
        ad = hipconv.act_dtype()
        shortcut = None
        if y1 is None and self.downsample is not None and hipconv.use_siblings(self.conv1, self.downsample[0], x):
            # first block of a stage: conv1 and the projection shortcut read the same map -- one launch over both sets of output channels
            y, shortcut = hipconv.conv_siblings(self.conv1, self.downsample[0], x)
        else:
            y = hipconv.conv(self.conv1, x, relu=True, out_dtype=torch.float32 if self.deformable else ad) if y1 is None else y1
        if self.deformable:
            off = hipconv.conv(self.conv2_offset, y)
            y, y_in = hipconv.dcn(self.conv2, y, off, relu=True), y
            hipconv._trace('dcn', module=self.conv2, xs=[y_in], offsets=[off], outs=[y], relu=True, 
                           form='dcn_fused' + (' bf16' if hipconv.ops.dcn_precision() == 'bf16' else ''))
        else:
            y = hipconv.conv(self.conv2, y, relu=True, out_dtype=ad, winograd='f2x2' if self.feeds_deformable else True)
        if shortcut is None:
            shortcut = x if self.downsample is None else hipconv.conv(self.downsample[0], x, out_dtype=ad)
        if nxt is not None and isinstance(nxt.bn1, nn.Identity) and hipconv.use_pair(self.conv3, nxt.conv1, y, shortcut):
            return hipconv.conv_pair(self.conv3, nxt.conv1, y, shortcut)
        return hipconv.conv(self.conv3, y, relu=True, residual=shortcut, out_dtype=ad), None


class Bottleneck(_Block):
    deformable = False


class DCNBottleneck(_Block):
    deformable = True


class conv1(nn.Module):
    """Stem: 7x7/2 conv + frozen BN + ReLU + 3x3/2 max-pool (always frozen)."""

    def __init__(self, requires_grad=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        if not requires_grad:
            self.eval()
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        if isinstance(self.bn1, nn.Identity) and hipconv.use_stem_pool(self.conv1, x):   # BN folded: conv + bias + ReLU + pool, one launch
            return hipconv.stem_pool(self.conv1, x)
        if isinstance(self.bn1, nn.Identity):  # BN folded: 7x7 conv + bias + ReLU on the MFMA kernel (NHWC4 image)
            return nn.functional.max_pool2d(hipconv.conv_stem(self.conv1, x, relu=True), 3, stride=2, padding=1)
        return nn.functional.max_pool2d(torch.relu_(self.bn1(self.conv1(x))), 3, stride=2, padding=1)


class res_block(nn.Module):
    """One stage: `blocks` bottlenecks under `.layers`; the first one carries the stride and a projection."""

    def __init__(self, planes, blocks, block=Bottleneck, stride=1, dilation=1, fix_bn=True, with_dpyramid=False):
        super().__init__()
        width_in = planes if planes == 64 else planes * 2
        width_out = planes * block.expansion
        proj = None
        if stride != 1 or width_in != width_out:
            proj = nn.Sequential(nn.Conv2d(width_in, width_out, 1, stride=stride, bias=False),
                                 _frozen_bn(width_out) if fix_bn else nn.BatchNorm2d(width_out))
        kinds = [block] * blocks
        if with_dpyramid:
            kinds[-1] = DCNBottleneck
        mods = [kinds[0](width_in, planes, stride, dilation, proj, fix_bn)]
        mods += [k(width_out, planes, dilation=dilation, fix_bn=fix_bn) for k in kinds[1:]]
        self.layers = nn.Sequential(*mods)

    def forward(self, x):
        blocks = list(self.layers)
        if not all(isinstance(b.bn1, nn.Identity) for b in blocks):
            return self.layers(x)
        y1 = None
        for i, blk in enumerate(blocks):
            x, y1 = blk.forward_chain(x, y1, blocks[i + 1] if i + 1 < len(blocks) else None)
        return x


class ResNetBackbone(nn.Module):
    """conv1 + res2..res5; returns the four stage outputs (strides 4, 8, 16, 32)."""

    def __init__(self, blocks):
        super().__init__()
        net = config.network
        self.fix_bn, self.freeze_at = net.backbone_fix_bn, net.backbone_freeze_at
        dconv_from = net.backbone_with_dconv           # DCN blocks in stage >= this index (100 = none)
        r5 = dict(stride=1, dilation=2) if net.backbone_with_dilation else dict(stride=2, dilation=1)
        self.conv1 = conv1(requires_grad=False)
        self.res2 = res_block(64, blocks[0], fix_bn=self.fix_bn)
        for idx, (planes, n, kw) in enumerate([(128, blocks[1], dict(stride=2, with_dpyramid=net.backbone_with_dpyramid)),
                                               (256, blocks[2], dict(stride=2, with_dpyramid=net.backbone_with_dpyramid)),
                                               (512, blocks[3], r5)], start=3):
            kind = DCNBottleneck if dconv_from <= idx else Bottleneck
            setattr(self, 'res%d' % idx, res_block(planes, n, block=kind, fix_bn=self.fix_bn, **kw))
        blocks_in_order = [b for name in ('res2', 'res3', 'res4', 'res5') for b in getattr(self, name).layers]
        last_dcn = max([i for i, b in enumerate(blocks_in_order) if b.deformable], default=-1)
        for b in blocks_in_order[:last_dcn + 1]:
            b.feeds_deformable = not b.deformable

    def forward(self, x):
        feats = []
        x = self.conv1(x)
        for name in ('res2', 'res3', 'res4', 'res5'):
            x = getattr(self, name)(x)
            feats.append(x)
        return tuple(feats)


# ------------------------------------------------------------------ inference re-parameterisation
def _fold(conv, bn):
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        conv.weight.mul_(scale.view(-1, 1, 1, 1))
        if conv.bias is None:
            conv.bias = nn.Parameter(shift.clone(), requires_grad=False)
        else:
            conv.bias.mul_(scale).add_(shift)


def fold_frozen_bn(module):
    """Fold each frozen BatchNorm2d into the conv / deformable conv in front of it (w' = w*s, b' = b*s + t)
    and replace it by nn.Identity. Call after the weights are loaded; state-dict keys of the folded BNs
    disappear, so checkpoints must be loaded first."""
    for m in module.modules():
        if isinstance(m, _Block):
            for c, b in (('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3')):
                if isinstance(getattr(m, b), nn.BatchNorm2d):
                    _fold(getattr(m, c), getattr(m, b))
                    setattr(m, b, nn.Identity())
            if m.downsample is not None and isinstance(m.downsample[1], nn.BatchNorm2d):
                _fold(m.downsample[0], m.downsample[1])
                m.downsample[1] = nn.Identity()
        elif isinstance(m, conv1) and isinstance(m.bn1, nn.BatchNorm2d):
            _fold(m.conv1, m.bn1)
            m.bn1 = nn.Identity()
    return module


# ------------------------------------------------------------------ checkpoint loading
_TORCHVISION_STAGE = {'layer1': 'res2', 'layer2': 'res3', 'layer3': 'res4', 'layer4': 'res5'}


class resnet_rcnn(nn.Module):
    """Tolerant state-dict loader with the reference's key mapping (upsnet/models/resnet.py:210-299):
    `resume=True` strips a DataParallel 'module.' prefix; otherwise torchvision/caffe ResNet keys
    (conv1/bn1/layerN) are mapped onto `resnet_backbone.*`. Shape mismatches and unknown keys warn
    instead of raising. (The reference's COCO->Cityscapes class-head surgery is a fine-tuning aid and
    is not part of the inference path.)"""

    def name_mapping(self, name, resume=False):
        if resume:
            return name[len('module.'):] if name.startswith('module.') else name
        if name.startswith(('conv1', 'bn1')):
            return 'resnet_backbone.conv1.' + name
        for tv, ours in _TORCHVISION_STAGE.items():
            name = name.replace(tv, 'resnet_backbone.%s.layers' % ours)
        return name

    def load_state_dict(self, state_dict, resume=False, strict=False):
        own = self.state_dict()
        mapped = {self.name_mapping(k, resume): v for k, v in state_dict.items()}
        for key, val in mapped.items():
            if key not in own:
                warnings.warn('unexpected key "%s" in state_dict' % key)
            elif own[key].shape != val.shape:
                warnings.warn('shape mismatch for %s: model %s vs checkpoint %s' % (key, tuple(own[key].shape), tuple(val.shape)))
            else:
                own[key].copy_(val.data if isinstance(val, nn.Parameter) else val)
        missing = sorted(set(own) - set(mapped))
        if missing:
            warnings.warn('missing keys in state_dict: %s' % missing)
