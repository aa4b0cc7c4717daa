"""ORACLE (test infrastructure, not product code): the camera branch of the multimodal frame (BASELINE configs[4]).

Two parts.

1. Restatements of the two third-party image trunks the reference imports and this image does not have:

   * ``EfficientNet`` -- the ``efficientnet_pytorch`` package (lukemelas/EfficientNet-PyTorch, v0.7.x), model
     "efficientnet-b0": stem Conv3x3/s2 + BN + swish, sixteen MBConv blocks (expand 1x1 -> depthwise k x k -> squeeze-excite
     -> project 1x1, identity skip when stride 1 and equal widths), then the classifier head the reference never calls.
     The convolutions are that package's ``Conv2dStaticSamePadding``: TensorFlow "same" padding computed ONCE from the
     package's nominal 224 x 224 input (so an odd-sized map is padded as its 224-derived size says, not as its own size
     would), BatchNorm eps 1e-3 / momentum 0.01.  Only the attributes the reference touches are kept, under the package's
     names (``_conv_stem``, ``_bn0``, ``_blocks``, ``_swish``, ``_global_params.drop_connect_rate``, ``from_pretrained``),
     so that ``lss_submodule.CamEncode`` (models/sub_modules/lss_submodule.py:50-189) runs on it unchanged and has the
     package's state_dict keys.
   * ``resnet18`` -- ``torchvision.models.resnet.resnet18`` (BasicBlock x [2, 2, 2, 2]); ``BevEncode``
     (lss_submodule.py:312-350) takes ``bn1``, ``relu``, ``layer1..3`` from it.

   **Trunk parity unpinned**: neither package can be imported here, so these two classes are written from the published
   definitions, and nothing checks them against the packages themselves.  Everything AROUND them (CamEncode, Up, BevEncode,
   LiftSplatShootEncoder.forward, fuse_bev) is the reference's own code run by tools/gen_golden.py with these classes
   registered in place of the missing packages; the fixtures ``cam_*.npz`` / ``w2c_cam_*.npz`` hold its outputs.

2. A functional restatement of the reference's camera encoder on a plain state_dict (``cam_encode``, ``bev_encode``,
   ``lss_encoder_forward``), citing what it follows; tests compare it with the fixtures (CPU) and the HIP path with it (GPU).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lss_oracle as lo

# ------------------------------------------------------------------------------------------------------------------
# efficientnet_pytorch restated
# ------------------------------------------------------------------------------------------------------------------
# (repeats, kernel, stride, expand, in, out, se_ratio) of efficientnet-b0 (width 1.0, depth 1.0)
B0_STAGES = ((1, 3, 1, 1, 32, 16, 0.25), (2, 3, 2, 6, 16, 24, 0.25), (2, 5, 2, 6, 24, 40, 0.25), (3, 3, 2, 6, 40, 80, 0.25),
             (3, 5, 1, 6, 80, 112, 0.25), (4, 5, 2, 6, 112, 192, 0.25), (1, 3, 1, 6, 192, 320, 0.25))
BN_EPS, BN_MOM, DROP_CONNECT, NOMINAL_SIZE = 1e-3, 0.01, 0.2, 224


def same_pad(size, k, s):
    """(before, after) zero padding of TF 'same' for a nominal input ``size``: total = max((ceil(size/s)-1)*s + k - size, 0)."""
    total = max((math.ceil(size / s) - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def b0_block_table():
    """Per MBConv block of B0: dict(cin, cout, k, s, expand, se, pad=(top/left, bottom/right) of the depthwise conv)."""
    rows, size = [], math.ceil(NOMINAL_SIZE / 2)          # nominal map size after the stem
    for rep, k, s, e, ci, co, se in B0_STAGES:
        for r in range(rep):
            st, cin = (s, ci) if r == 0 else (1, co)
            rows.append(dict(cin=cin, cout=co, k=k, s=st, expand=e, se=max(1, int(cin * se)), pad=same_pad(size, k, st)))
            size = math.ceil(size / st)
    return rows


class SamePadConv2d(nn.Conv2d):
    """Conv2dStaticSamePadding: zero padding fixed at construction from a nominal image size."""

    def __init__(self, cin, cout, kernel_size, stride=1, image_size=None, **kw):
        super().__init__(cin, cout, kernel_size, stride, **kw)
        k, s = self.kernel_size[0], self.stride[0]
        (t, b), (l, r) = same_pad(image_size, k, s), same_pad(image_size, k, s)
        self.static_padding = nn.ZeroPad2d((l, r, t, b)) if (t + b + l + r) > 0 else nn.Identity()

    def forward(self, x):
        return F.conv2d(self.static_padding(x), self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class MBConvBlock(nn.Module):
    def __init__(self, row, size):
        super().__init__()
        self.row = row
        cin, mid = row["cin"], row["cin"] * row["expand"]
        if row["expand"] != 1:
            self._expand_conv = SamePadConv2d(cin, mid, 1, image_size=size, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = SamePadConv2d(mid, mid, row["k"], row["s"], image_size=size, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        self._se_reduce = SamePadConv2d(mid, row["se"], 1, image_size=1)
        self._se_expand = SamePadConv2d(row["se"], mid, 1, image_size=1)
        self._project_conv = SamePadConv2d(mid, row["cout"], 1, image_size=1, bias=False)
        self._bn2 = nn.BatchNorm2d(row["cout"], momentum=BN_MOM, eps=BN_EPS)
        self._swish = Swish()

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.row["expand"] != 1:
            x = self._swish(self._bn0(self._expand_conv(x)))
        x = self._swish(self._bn1(self._depthwise_conv(x)))
        g = self._se_expand(self._swish(self._se_reduce(F.adaptive_avg_pool2d(x, 1))))
        x = torch.sigmoid(g) * x
        x = self._bn2(self._project_conv(x))
        if self.row["s"] == 1 and self.row["cin"] == self.row["cout"]:
            if drop_connect_rate and self.training:      # stochastic depth: training only
                keep = 1 - drop_connect_rate
                mask = torch.floor(keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device))
                x = x / keep * mask
            x = x + inputs
        return x


class EfficientNet(nn.Module):
    def __init__(self):
        super().__init__()
        self._global_params = SimpleNamespace(drop_connect_rate=DROP_CONNECT, batch_norm_epsilon=BN_EPS, image_size=NOMINAL_SIZE)
        self._conv_stem = SamePadConv2d(3, 32, 3, 2, image_size=NOMINAL_SIZE, bias=False)
        self._bn0 = nn.BatchNorm2d(32, momentum=BN_MOM, eps=BN_EPS)
        size = math.ceil(NOMINAL_SIZE / 2)
        blocks = []
        for row in b0_block_table():
            blocks.append(MBConvBlock(row, size))
            size = math.ceil(size / row["s"])
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = SamePadConv2d(320, 1280, 1, image_size=size, bias=False)     # classifier head: parameters only
        self._bn1 = nn.BatchNorm2d(1280, momentum=BN_MOM, eps=BN_EPS)
        self._fc = nn.Linear(1280, 1000)
        self._swish = Swish()

    @classmethod
    def from_pretrained(cls, name, **kw):
        if name != "efficientnet-b0":
            raise NotImplementedError(name)
        return cls()        # no weights here: the caller loads a state_dict


# ------------------------------------------------------------------------------------------------------------------
# torchvision.models.resnet.resnet18 restated
# ------------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + idt)


class ResNet18(nn.Module):
    def __init__(self, zero_init_residual=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths, cin = (64, 128, 256, 512), 64
        for i, c in enumerate(widths):
            setattr(self, f"layer{i + 1}", nn.Sequential(BasicBlock(cin, c, 1 if i == 0 else 2), BasicBlock(c, c, 1)))
            cin = c
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)


def resnet18(pretrained=False, zero_init_residual=False, **kw):
    return ResNet18(zero_init_residual)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (expansion 4, stride on the 3x3), restated from its published definition."""

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != 4 * planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride, bias=False), nn.BatchNorm2d(4 * planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class ResNet101Front(nn.Module):
    """The part of torchvision's resnet101 CamEncode_Resnet101 keeps (lss_submodule.py:206-216): conv1, bn1, maxpool, layer1 (3 bottlenecks,
    planes 64), layer2 (4 bottlenecks, planes 128, stride 2).  layer3 / layer4 / fc are never registered there, so they are not built."""

    def __init__(self, zero_init_residual=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1), Bottleneck(256, 64, 1), Bottleneck(256, 64, 1))
        self.layer2 = nn.Sequential(Bottleneck(256, 128, 2), *[Bottleneck(512, 128, 1) for _ in range(3)])
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)


def resnet101(pretrained=False, zero_init_residual=False, **kw):      # lss_submodule.py:206
    return ResNet101Front(zero_init_residual)


# ------------------------------------------------------------------------------------------------------------------
# functional restatement of the reference's camera encoder on a state_dict
# ------------------------------------------------------------------------------------------------------------------
def _bn(sd, p, x, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _swish(x):
    return x * torch.sigmoid(x)


def effnet_features(sd, p, x):
    """CamEncode.get_eff_features' trunk walk (lss_submodule.py:118-146): -> the maps at strides 8, 16, 32 (reduction_3/4/5).
    ``p`` = prefix of the trunk's keys (e.g. "veh_models.0.camencode.trunk.")."""
    t, b = same_pad(NOMINAL_SIZE, 3, 2)
    x = _swish(_bn(sd, p + "_bn0", F.conv2d(F.pad(x, (t, b, t, b)), sd[p + "_conv_stem.weight"], None, 2), BN_EPS))
    ends, prev = [], x
    for i, row in enumerate(b0_block_table()):
        q = f"{p}_blocks.{i}."
        inp = x
        mid = row["cin"] * row["expand"]
        if row["expand"] != 1:
            x = _swish(_bn(sd, q + "_bn0", F.conv2d(x, sd[q + "_expand_conv.weight"]), BN_EPS))
        a, c = row["pad"]
        x = _swish(_bn(sd, q + "_bn1", F.conv2d(F.pad(x, (a, c, a, c)), sd[q + "_depthwise_conv.weight"], None, row["s"], 0, 1, mid), BN_EPS))
        g = F.conv2d(x.mean((2, 3), keepdim=True), sd[q + "_se_reduce.weight"], sd[q + "_se_reduce.bias"])
        g = F.conv2d(_swish(g), sd[q + "_se_expand.weight"], sd[q + "_se_expand.bias"])
        x = torch.sigmoid(g) * x
        x = _bn(sd, q + "_bn2", F.conv2d(x, sd[q + "_project_conv.weight"]), BN_EPS)
        if row["s"] == 1 and row["cin"] == row["cout"]:
            x = x + inp
        if prev.shape[2] > x.shape[2]:
            ends.append(prev)
        prev = x
    ends.append(x)
    return ends[2], ends[3], ends[4]


def up_block(sd, p, x1, x2, scale):
    """Up.forward (lss_submodule.py:22-47): bilinear x ``scale`` (align_corners), pad to x2, cat([x2, x1]), two Conv3x3+BN+ReLU."""
    x1 = F.interpolate(x1, scale_factor=scale, mode="bilinear", align_corners=True)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    x = torch.cat([x2, x1], 1)
    x = F.relu(_bn(sd, p + "conv.1", F.conv2d(x, sd[p + "conv.0.weight"], None, 1, 1), 1e-5))
    return F.relu(_bn(sd, p + "conv.4", F.conv2d(x, sd[p + "conv.3.weight"], None, 1, 1), 1e-5))


def bin_depths(depth, mode, dmin, dmax, nbins, target):
    """utils/camera_utils.py:247-298 (UD / LID), same tensor ops: -> (int64 indices, valid mask or None)."""
    if mode == "UD":
        idx = (depth - dmin) / ((dmax - dmin) / nbins)
    elif mode == "LID":
        idx = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth - dmin) / (2 * (dmax - dmin) / (nbins * (1 + nbins))))
    else:
        raise NotImplementedError(mode)
    bad = (idx < 0) | (idx >= nbins) | (~torch.isfinite(idx))
    idx = idx.clone()
    idx[idx < 0] = 0
    idx[idx >= nbins] = nbins - 1
    idx[~torch.isfinite(idx)] = nbins - 1
    return idx.type(torch.int64), (None if target else ~bad)


def cam_encode(sd, p, imgs, cam_args, training=False):
    """CamEncode.forward (lss_submodule.py:148-189), downsample 8 or 16: imgs (BN, 4, H, W) -> (image features (BN, C, fH, fW),
    depth distribution (BN, D, fH, fW) float)."""
    ds = cam_args["img_downsample"]
    if ds not in (8, 16):
        raise NotImplementedError("img_downsample 8 or 16 (lss_submodule.py:72-75)")
    dmin, dmax, nb = cam_args["grid_conf"]["ddiscr"]
    if cam_args.get("camera_encoder", "EfficientNet") == "Resnet101":
        f = resnet101_features(sd, p, imgs[:, :3])                                  # CamEncode_Resnet101.forward :280-310 (512 channels)
    else:
        r3, r4, r5 = effnet_features(sd, p + "trunk.", imgs[:, :3])
        f = up_block(sd, p + "up1.", r5, r4, 2)
        if ds == 8:                                                                 # :152-153
            f = up_block(sd, p + "up2.", f, r3, 2)
    x_img = F.conv2d(f, sd[p + "image_head.weight"], sd[p + "image_head.bias"])
    if cam_args["use_depth_gt"]:
        d = torch.clamp(imgs[:, 3], max=dmax)                                       # :103 (clamp_max_)
        idx, mask = bin_depths(d, cam_args["grid_conf"]["mode"], dmin, dmax, nb, target=training)
        idx = idx[:, ds // 2::ds, ds // 2::ds]
        dist = F.one_hot(idx, nb).permute(0, 3, 1, 2)
        if mask is not None:
            dist = dist * mask[:, ds // 2::ds, ds // 2::ds].unsqueeze(1)
        return x_img, dist.float()
    logit = F.conv2d(f, sd[p + "depth_head.weight"], sd[p + "depth_head.bias"])
    return x_img, F.softmax(logit, 1)


def bottleneck(sd, p, x, stride):
    idt = x
    if (p + "downsample.0.weight") in sd:
        idt = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), 1e-5)
    y = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"]), 1e-5))
    y = F.relu(_bn(sd, p + "bn2", F.conv2d(y, sd[p + "conv2.weight"], stride=stride, padding=1), 1e-5))
    return F.relu(_bn(sd, p + "bn3", F.conv2d(y, sd[p + "conv3.weight"]), 1e-5) + idt)


def resnet101_features(sd, p, x):
    """CamEncode_Resnet101.resnet101_forward (:262-270): conv1 7x7/2 + bn1 + relu + maxpool 3x3/2 + layer1 + layer2 -> (BN, 512, H/8, W/8)."""
    x = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3), 1e-5))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (nb, stride) in enumerate(((3, 1), (4, 2)), 1):
        for bi in range(nb):
            x = bottleneck(sd, f"{p}layer{li}.{bi}.", x, stride if bi == 0 else 1)
    return x


def basic_block(sd, p, x, stride):
    idt = x
    if (p + "downsample.0.weight") in sd:
        idt = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride), 1e-5)
    y = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1), 1e-5))
    y = _bn(sd, p + "bn2", F.conv2d(y, sd[p + "conv2.weight"], None, 1, 1), 1e-5)
    return F.relu(y + idt)


def bev_encode(sd, p, x):
    """BevEncode.forward (lss_submodule.py:335-350)."""
    x = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3), 1e-5))
    x1 = basic_block(sd, p + "layer1.1.", basic_block(sd, p + "layer1.0.", x, 1), 1)
    x = basic_block(sd, p + "layer2.1.", basic_block(sd, p + "layer2.0.", x1, 2), 1)
    x = basic_block(sd, p + "layer3.1.", basic_block(sd, p + "layer3.0.", x, 2), 1)
    x = up_block(sd, p + "up1.", x, x1, 4)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.relu(_bn(sd, p + "up2.2", F.conv2d(x, sd[p + "up2.1.weight"], None, 1, 1), 1e-5))
    return F.conv2d(x, sd[p + "up2.4.weight"], sd[p + "up2.4.bias"])


def lss_encoder_forward(sd, p, cam_inputs, cam_args, training=False, trace=None):
    """LiftSplatShootEncoder.forward (airv2x_encoder.py:309-336): cam_inputs = batch_merged_cam_inputs of one agent type
    (imgs (B, N, 4, H, W), rots, trans, intrinsics, post_rots, post_trans) -> spatial_features (B, bevout, ny, nx)."""
    imgs = cam_inputs["imgs"].float()
    B, N = imgs.shape[:2]
    g = cam_args["grid_conf"]
    dx, bx, nx = lo.gen_dx_bx(g["xbound"], g["ybound"], g["zbound"])
    frustum = lo.create_frustum(g, cam_args["data_aug_conf"], cam_args["img_downsample"])
    geom = lo.get_geometry(frustum, cam_inputs["rots"].float(), cam_inputs["trans"].float(), cam_inputs["intrinsics"].float(),
                           cam_inputs["post_rots"].float(), cam_inputs["post_trans"].float())
    x_img, dist = cam_encode(sd, p + "camencode.", imgs.reshape(B * N, *imgs.shape[2:]), cam_args, training)
    x = dist.unsqueeze(1) * x_img.unsqueeze(2)                                                    # (BN, C, D, fH, fW)
    x = x.view(B, N, *x.shape[1:]).permute(0, 1, 3, 4, 5, 2)
    bev = lo.voxel_pooling(geom, x, dx, bx, nx)
    if trace is not None:
        trace.update(x_img=x_img, dist=dist, pooled=bev)
    return bev_encode(sd, p + "bevencode.", bev)
