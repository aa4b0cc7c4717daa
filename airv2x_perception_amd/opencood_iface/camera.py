"""Camera encoder of the multimodal frame on the device (BASELINE configs[4]).

``LiftSplatShootEncoder.forward`` (models/common_modules/airv2x_encoder.py:309-336) for one agent type:

    imgs (B, N, 4, H, W) ─► CamEncode (models/sub_modules/lss_submodule.py:50-189)
                               EfficientNet-B0 trunk (efficientnet_pytorch, walked by get_eff_features :118-146):
                                 stem [av2x_cam_stem] ─► 16 x MBConv { expand 1x1 [av2x_conv2d, swish epilogue] ─► depthwise k x k
                                 [av2x_dwconv2d] ─► squeeze-excite [av2x_squeeze_excite] ─► project 1x1 (+ skip) [av2x_conv2d_res] }
                               Up(320+112 -> 256), Up(256+40 -> 256)  [av2x_resize_bilinear into the concat buffer, Winograd 3x3 convs]
                               image_head 1x1 ─► features (BN, fH, fW, C);  depth: ground-truth plane binned on the fly, or depth_head + softmax
                           ─► lift + splat [av2x_lss_lift_pool: the (B,N,D,fH,fW,C) volume is never written] ─► (B, ny, nx, C)
                           ─► BevEncode (:312-350): 7x7/s2 stem, ResNet-18 layer1-3 (BasicBlocks: residual-then-ReLU epilogue), Up x4, Up x2,
                              3x3 + 1x1  ─► spatial_features (B, ny, nx, bevout) NHWC

All tensors NHWC fp32 in the engine's workspace pool.  Channel counts that are not multiples of 32 (EfficientNet's 16 / 24 / 40 / 80 /
112 / 144 / 240) are zero-padded to the next multiple once, in the packed weights: the padded channels compute exact zeros, so no
kernel needs a ragged-K path.  The trunk packages (efficientnet_pytorch, torchvision) are absent from this image: the layer
structure follows their published definitions (oracle/camera_oracle.py restates them; "trunk parity unpinned" there).
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int32, c_void_p

import torch

from .. import _lib
from ..synth import effnet_b0_blocks
from .lss import create_frustum, gen_dx_bx
from .packing import fold_bn, pack_conv_weight, round_up

EFF_EPS = 1e-3      # efficientnet_pytorch batch_norm_epsilon
TV_EPS = 1e-5       # nn.BatchNorm2d default (Up, BevEncode, torchvision BasicBlock)
SWISH, RELU_AFTER_RES = 6, 5


def _pad32(c):
    return round_up(c, 32)


def _pad_vec(v, n, fill):
    out = torch.full((n,), fill, dtype=torch.float32)
    out[: v.numel()] = v.detach().float().cpu().reshape(-1)
    return out


def _place_cin(w, segments, cin_p):
    """(cout, cin, k, k) -> (cout, cin_p, k, k): input-channel runs [(src0, src1, dst0)] copied, zeros elsewhere."""
    w = w.detach().float().cpu()
    out = torch.zeros(w.shape[0], cin_p, w.shape[2], w.shape[3])
    for s0, s1, d0 in segments:
        out[:, d0:d0 + (s1 - s0)] = w[:, s0:s1]
    return out


def _cam_calibration_on_host(ci):
    """rots, trans, intrinsics, post_rots, post_trans of a camera batch as fp32 CPU tensors.  Device-resident inputs come back in ONE
    read-back (concatenated on the device) instead of five: every .to("cpu") is a stream synchronise of its own."""
    keys = ("rots", "trans", "intrinsics", "post_rots", "post_trans")
    ts = [ci[k].detach() for k in keys]
    if all(t.device.type == "cpu" for t in ts):
        return [t.to(torch.float32) for t in ts]
    B, N = ts[1].shape[:2]
    dev = next(t.device for t in ts if t.device.type != "cpu")
    flat = torch.cat([t.to(dev, torch.float32).reshape(B * N, -1) for t in ts], 1).to("cpu")
    return [flat[:, 0:9].reshape(B, N, 3, 3), flat[:, 9:12].reshape(B, N, 3), flat[:, 12:21].reshape(B, N, 3, 3), flat[:, 21:30].reshape(B, N, 3, 3),
            flat[:, 30:33].reshape(B, N, 3)]


def _check_downsample(cam_args):
    """img_downsample as the reference's CamEncode takes it (lss_submodule.py:72-75, 152-153): 8 = up1 + up2 on the stride-8 EfficientNet
    endpoint (every shipped AirV2X config); 16 = up1 only, features at stride 16 (``up2`` does not exist in the state_dict then).  Any other
    value gives a feature map that does not match the frustum in the reference either; CamEncode_Resnet101 ends at its stride-8 layer2."""
    ds = int(cam_args["img_downsample"])
    if ds not in (8, 16) or (ds == 16 and cam_args["camera_encoder"] == "Resnet101"):
        raise NotImplementedError(f"img_downsample {ds} with {cam_args['camera_encoder']}: 8 (EfficientNet / Resnet101) or 16 (EfficientNet)")


class CameraGeometry:
    """The weight-free part of one agent type's LiftSplatShootEncoder: frustum, BEV grid, depth bins and the per-camera matrices
    (airv2x_encoder.py:31-167).  The training forward (train_camera.py) uses it on its own; CameraEncoder builds the same values."""

    def __init__(self, cam_args, device):
        self.cfg, self.device = cam_args, device
        if cam_args["camera_encoder"] not in ("EfficientNet", "Resnet101"):
            raise NotImplementedError(f"camera_encoder {cam_args['camera_encoder']!r}: EfficientNet or Resnet101 (airv2x_encoder.py:67-86)")
        _check_downsample(cam_args)
        g = cam_args["grid_conf"]
        self.dx, self.bx, self.nx = gen_dx_bx(g["xbound"], g["ybound"], g["zbound"])
        self.ds = int(cam_args["img_downsample"])
        self.C = int(cam_args["img_features"])
        self.outC = int(cam_args["bevout_feature"])
        fr = create_frustum(g, cam_args["data_aug_conf"], self.ds)
        self.D, self.fH, self.fW = [int(v) for v in fr.shape[:3]]
        self.frustum = fr.contiguous().view(-1, 3).to(device)
        lo = self.bx - self.dx / 2.0
        self._lo = (c_float * 3)(*[float(v) for v in lo])
        self._dx = (c_float * 3)(*[float(v) for v in self.dx])
        self._nx = (c_int32 * 3)(*[int(v) for v in self.nx])
        dmin, dmax, nb = g["ddiscr"]
        if g["mode"] == "UD":
            bin_size, self.depth_mode = (dmax - dmin) / nb, 0
        elif g["mode"] == "LID":
            bin_size, self.depth_mode = 2 * (dmax - dmin) / (nb * (1 + nb)), 1
        else:
            raise NotImplementedError(f"depth discretisation {g['mode']} (UD / LID)")
        self._depth3 = (c_float * 3)(float(dmin), float(dmax), float(bin_size))
        self.nbins = int(nb)
        self.use_gt = bool(cam_args["use_depth_gt"])
        if int(self.nx[2]) != 1:
            raise NotImplementedError("camera BEV grid with more than one z cell")

    def _cam_params(self, ci):
        """(B*N, 24) device rows [inverse(post_rots) | post_trans | rots @ inverse(intrins) | trans] (airv2x_encoder.py:147-166)."""
        rots, trans, intr, prot, ptr = _cam_calibration_on_host(ci)
        B, N = trans.shape[:2]
        rows = torch.cat([torch.inverse(prot).reshape(B * N, 9), ptr.reshape(B * N, 3), rots.matmul(torch.inverse(intr)).reshape(B * N, 9),
                          trans.reshape(B * N, 3)], 1).contiguous()
        return rows.to(self.device)


class CameraEncoder:
    """Packed weights + launch sequence of one agent type's LiftSplatShootEncoder.  ``eng``: the owning engine (workspace pool,
    conv launcher with its tile tuner, stream)."""

    IMG_CHUNK = 32      # images per trunk pass / agents per BevEncode pass: keeps every map inside the 2 GiB buffer-descriptor window
    BEV_CHUNK = 8

    def __init__(self, eng, cam_args, sd, prefix, tag):
        from .engine import ConvLayer
        self.eng, self.tag, self.cfg = eng, tag, cam_args
        if cam_args["camera_encoder"] not in ("EfficientNet", "Resnet101"):
            raise NotImplementedError(f"camera_encoder {cam_args['camera_encoder']!r}: EfficientNet or Resnet101 (airv2x_encoder.py:67-86)")
        self.resnet = cam_args["camera_encoder"] == "Resnet101"
        _check_downsample(cam_args)
        self.lib = eng.lib
        up = eng._up
        g = cam_args["grid_conf"]
        self.dx, self.bx, self.nx = gen_dx_bx(g["xbound"], g["ybound"], g["zbound"])
        self.ds = int(cam_args["img_downsample"])
        self.C = int(cam_args["img_features"])
        self.outC = int(cam_args["bevout_feature"])
        fr = create_frustum(g, cam_args["data_aug_conf"], self.ds)
        self.D, self.fH, self.fW = [int(v) for v in fr.shape[:3]]
        self.frustum = up(fr.contiguous().view(-1, 3))
        lo = self.bx - self.dx / 2.0
        self._lo = (c_float * 3)(*[float(v) for v in lo])
        self._dx = (c_float * 3)(*[float(v) for v in self.dx])
        self._nx = (c_int32 * 3)(*[int(v) for v in self.nx])
        dmin, dmax, nb = g["ddiscr"]
        if g["mode"] == "UD":
            bin_size, self.depth_mode = (dmax - dmin) / nb, 0
        elif g["mode"] == "LID":
            bin_size, self.depth_mode = 2 * (dmax - dmin) / (nb * (1 + nb)), 1
        else:
            raise NotImplementedError(f"depth discretisation {g['mode']} (UD / LID)")
        self._depth3 = (c_float * 3)(float(dmin), float(dmax), float(bin_size))
        self.nbins = int(nb)
        self.use_gt = bool(cam_args["use_depth_gt"])
        if int(self.nx[2]) != 1:
            raise NotImplementedError("camera BEV grid with more than one z cell")

        def conv(w, scale, shift, relu, stride=1, pad=0, cin_p=None, cout_p=None):
            """Conv2d weight (cout, cin, k, k) (+ folded BN) -> ConvLayer on zero-padded channel counts."""
            w = w.detach().float().cpu()
            cout, cin, k, _ = w.shape
            cin_p = cin_p or cin
            if cin_p != cin:
                w = _place_cin(w, [(0, cin, 0)], cin_p)
            real = cout
            if cout_p and cout_p != cout:     # padded outputs are REAL columns of zeros (scale 1, shift 0): the buffer holds exact zeros
                w = torch.cat([w, torch.zeros(cout_p - cout, *w.shape[1:])], 0)
                scale = _pad_vec(scale, cout_p, 1.0) if scale is not None else None
                shift = _pad_vec(shift, cout_p, 0.0)
                real = cout_p
            pw, coutp = pack_conv_weight(w)
            return ConvLayer(up(pw), up(scale) if scale is not None else None, up(shift.detach().float()), cin_p, real, coutp, k, stride, pad, relu)

        # ---- Up blocks of CamEncode: concat layouts [skip map (padded) | upsampled map]
        def up_block(pp, c_skip, c_up):
            cs_p = _pad32(c_skip)
            w0 = _place_cin(sd[pp + "conv.0.weight"], [(0, c_skip, 0), (c_skip, c_skip + c_up, cs_p)], cs_p + c_up)
            sa, ha = fold_bn(sd, pp + "conv.1", TV_EPS)
            sb, hb = fold_bn(sd, pp + "conv.4", TV_EPS)
            return {"cs_p": cs_p, "c_up": c_up, "c0": conv(w0, sa, ha, 1, pad=1), "c1": conv(sd[pp + "conv.3.weight"], sb, hb, 1, pad=1)}

        p = prefix + "camencode."
        t = p + "trunk."
        if self.resnet:
            # ---- CamEncode_Resnet101 (lss_submodule.py:191-310): conv1 7x7/2 (image channels padded to 16) + bn1 + relu, maxpool 3x3/2,
            # torchvision resnet101's layer1 (3 bottlenecks) and layer2 (4, stride 2) -> 512 channels at stride 8; BatchNorms folded
            from ..synth import RESNET101_LAYERS
            s0, h0 = fold_bn(sd, p + "bn1", TV_EPS)
            self.r_stem = conv(sd[p + "conv1.weight"], s0, h0, 1, stride=2, pad=3, cin_p=16)
            self.r_blocks = []
            for li, (planes, nb, stride) in enumerate(RESNET101_LAYERS, 1):
                for bi in range(nb):
                    q = f"{p}layer{li}.{bi}."
                    st_ = stride if bi == 0 else 1
                    b1, b2, b3 = (fold_bn(sd, q + f"bn{j}", TV_EPS) for j in (1, 2, 3))
                    blk = {"c1": conv(sd[q + "conv1.weight"], b1[0], b1[1], 1), "c2": conv(sd[q + "conv2.weight"], b2[0], b2[1], 1, stride=st_, pad=1),
                           "c3": conv(sd[q + "conv3.weight"], b3[0], b3[1], RELU_AFTER_RES), "down": None}
                    if (q + "downsample.0.weight") in sd:
                        sd_, hd = fold_bn(sd, q + "downsample.1", TV_EPS)
                        blk["down"] = conv(sd[q + "downsample.0.weight"], sd_, hd, 0, stride=st_)
                    self.r_blocks.append(blk)
        # ---- EfficientNet-B0 trunk
        if not self.resnet:
            self._build_effnet(sd, p, t, conv, up, up_block)
        self.image_head = conv(sd[p + "image_head.weight"], None, sd[p + "image_head.bias"], 0)
        self.depth_head = None if self.use_gt else conv(sd[p + "depth_head.weight"], None, sd[p + "depth_head.bias"], 0)
        self._build_bevencode(sd, prefix, conv, up_block)

    def _build_effnet(self, sd, p, t, conv, up, up_block):
        sc, sh = fold_bn(sd, t + "_bn0", EFF_EPS)
        self.stem = (up(sd[t + "_conv_stem.weight"].detach().float().cpu().permute(2, 3, 1, 0).reshape(27, 32).contiguous()), up(sc), up(sh))
        self.blocks = []
        for i, (cin, cout, k, st, e, se, pad) in enumerate(effnet_b0_blocks()):
            q, mid = f"{t}_blocks.{i}.", cin * e
            cin_p, mid_p, cout_p = _pad32(cin), _pad32(mid), _pad32(cout)
            b = {"k": k, "s": st, "pad": pad, "mid_p": mid_p, "cout_p": cout_p, "se": se, "skip": st == 1 and cin == cout, "expand": None}
            if e != 1:
                s0, h0 = fold_bn(sd, q + "_bn0", EFF_EPS)
                b["expand"] = conv(sd[q + "_expand_conv.weight"], s0, h0, SWISH, cin_p=cin_p, cout_p=mid_p)
            s1, h1 = fold_bn(sd, q + "_bn1", EFF_EPS)
            dw = torch.zeros(k * k, mid_p)
            dw[:, :mid] = sd[q + "_depthwise_conv.weight"].detach().float().cpu().reshape(mid, k * k).t()
            b["dw"] = (up(dw.contiguous()), up(_pad_vec(s1, mid_p, 1.0)), up(_pad_vec(h1, mid_p, 0.0)))
            wr = torch.zeros(se, mid_p)
            wr[:, :mid] = sd[q + "_se_reduce.weight"].detach().float().cpu().reshape(se, mid)
            we = torch.zeros(se, mid_p)                                                   # transposed: (c_se, c)
            we[:, :mid] = sd[q + "_se_expand.weight"].detach().float().cpu().reshape(mid, se).t()
            b["se_w"] = (up(wr.contiguous()), up(sd[q + "_se_reduce.bias"].detach().float()), up(we.contiguous()),
                         up(_pad_vec(sd[q + "_se_expand.bias"], mid_p, 0.0)))
            s2, h2 = fold_bn(sd, q + "_bn2", EFF_EPS)
            b["project"] = conv(sd[q + "_project_conv.weight"], s2, h2, 0, cin_p=mid_p, cout_p=cout_p)
            self.blocks.append(b)
        self.up1 = up_block(p + "up1.", 112, 320)
        self.up2 = up_block(p + "up2.", 40, 256) if int(self.cfg["img_downsample"]) == 8 else None    # lss_submodule.py:74-75

    def _build_bevencode(self, sd, prefix, conv, up_block):
        # ---- BevEncode
        b = prefix + "bevencode."
        s, h = fold_bn(sd, b + "bn1", TV_EPS)
        self.bev_stem = conv(sd[b + "conv1.weight"], s, h, 1, stride=2, pad=3)
        self.layers = []
        cin = 64
        for li, c in enumerate((64, 128, 256)):
            for bi in range(2):
                q = f"{b}layer{li + 1}.{bi}."
                stride = 2 if (li > 0 and bi == 0) else 1
                s1, h1 = fold_bn(sd, q + "bn1", TV_EPS)
                s2, h2 = fold_bn(sd, q + "bn2", TV_EPS)
                blk = {"c1": conv(sd[q + "conv1.weight"], s1, h1, 1, stride=stride, pad=1),
                       "c2": conv(sd[q + "conv2.weight"], s2, h2, RELU_AFTER_RES, pad=1), "down": None}
                if (q + "downsample.0.weight") in sd:
                    sd_, hd = fold_bn(sd, q + "downsample.1", TV_EPS)
                    blk["down"] = conv(sd[q + "downsample.0.weight"], sd_, hd, 0, stride=stride)
                self.layers.append(blk)
                cin = c
        self.bev_up1 = up_block(b + "up1.", 64, 256)
        s, h = fold_bn(sd, b + "up2.2", TV_EPS)
        self.bev_up2 = conv(sd[b + "up2.1.weight"], s, h, 1, pad=1)
        self.bev_out = conv(sd[b + "up2.4.weight"], None, sd[b + "up2.4.bias"], 0)

    # ------------------------------------------------------------------------------------------------------------ pieces
    def _cam_params(self, ci):
        """(B*N, 24) device rows [inverse(post_rots) | post_trans | rots @ inverse(intrins) | trans] (airv2x_encoder.py:147-166),
        computed on the host in fp32 with the reference's own torch calls, like the V2X-ViT correction matrices."""
        rots, trans, intr, prot, ptr = _cam_calibration_on_host(ci)
        B, N = trans.shape[:2]
        rows = torch.cat([torch.inverse(prot).reshape(B * N, 9), ptr.reshape(B * N, 3), rots.matmul(torch.inverse(intr)).reshape(B * N, 9),
                          trans.reshape(B * N, 3)], 1).contiguous()
        return rows.to(self.eng.device)

    def _mbconv(self, i, x, n, h, w, tag):
        e, lib, st, P = self.eng, self.lib, self.eng.stream(), lambda t: c_void_p(t.data_ptr())
        b = self.blocks[i]
        inp = x
        mid = b["mid_p"]
        if b["expand"] is not None:
            y = e.buf(f"cam_exp_{tag}", (n, h, w, mid))
            e.conv(b["expand"], x, n, h, w, y)
            x = y
        pa, pb = b["pad"]
        ho = (h + pa + pb - b["k"]) // b["s"] + 1
        wo = (w + pa + pb - b["k"]) // b["s"] + 1
        z = e.buf(f"cam_dw_{tag}", (n, ho, wo, mid))
        dw, dsc, dsh = b["dw"]
        _lib.check(lib.av2x_dwconv2d(P(x), n, h, w, mid, P(dw), P(dsc), P(dsh), b["k"], b["s"], pa, pa, ho, wo, SWISH, P(z), st), "av2x_dwconv2d")
        ws = e.buf(f"cam_se_{tag}", (int(lib.av2x_squeeze_excite_workspace_bytes(n, ho * wo, mid)) // 4,))
        wr, br, we, be = b["se_w"]
        _lib.check(lib.av2x_squeeze_excite(P(z), n, ho * wo, mid, P(wr), P(br), b["se"], P(we), P(be), P(ws), 1, st), "av2x_squeeze_excite")
        out = e.buf(f"cam_blk{i}_{tag}", (n, ho, wo, b["cout_p"]))
        e.conv(b["project"], z, n, ho, wo, out, residual=inp if b["skip"] else None)
        return out, ho, wo

    def _up(self, U, x1, h1, w1, x2, h2, w2, c2_tot, scale, n, tag):
        """Up.forward (:39-47): upsample x1 by ``scale``, pad to x2's size, concat [x2 | x1], two Conv3x3 + BN + ReLU."""
        e, lib, st, P = self.eng, self.lib, self.eng.stream(), lambda t: c_void_p(t.data_ptr())
        ctot = U["cs_p"] + U["c_up"]
        cat = e.buf(f"cam_cat_{tag}", (n, h2, w2, ctot))
        _lib.check(lib.av2x_resize_bilinear(P(x2), n, h2, w2, U["cs_p"], c2_tot, 0, h2, w2, 0, 0, h2, w2, P(cat), ctot, 0, st), "av2x_resize_bilinear")
        H2, W2 = h1 * scale, w1 * scale
        dy, dxx = h2 - H2, w2 - W2
        if dy < 0 or dxx < 0:
            raise NotImplementedError("Up: the upsampled map is larger than the skip map (negative F.pad)")
        _lib.check(lib.av2x_resize_bilinear(P(x1), n, h1, w1, U["c_up"], U["c_up"], 0, H2, W2, dy // 2, dxx // 2, h2, w2, P(cat), ctot, U["cs_p"], st),
                   "av2x_resize_bilinear")
        y0 = e.buf(f"cam_up0_{tag}", (n, h2, w2, U["c0"].cout))
        e.conv(U["c0"], cat, n, h2, w2, y0)
        y1 = e.buf(f"cam_up1_{tag}", (n, h2, w2, U["c1"].cout))
        e.conv(U["c1"], y0, n, h2, w2, y1)
        return y1

    def features(self, imgs, n, planes, H, W, tag, feat, prob, trace=None):
        """CamEncode.get_eff_features + heads on n images (device NCHW) -> feat (n,fH,fW,C), and (predicted depth) prob (n,fH,fW,D)."""
        e, lib, st, P = self.eng, self.lib, self.eng.stream(), lambda t: c_void_p(t.data_ptr())
        if self.resnet:
            return self._features_resnet(imgs, n, planes, H, W, tag, feat, prob, trace)
        h, w = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1          # static "same" padding of the 224-nominal stem: 0 before, 1 after
        x = e.buf(f"cam_stem_{tag}", (n, h, w, 32))
        sw, ssc, ssh = self.stem
        _lib.check(lib.av2x_cam_stem(P(imgs), n, planes, H, W, P(sw), P(ssc), P(ssh), 0, 0, h, w, P(x), st), "av2x_cam_stem")
        ends, prev = [], (x, h, w)
        for i in range(len(self.blocks)):
            x, ho, wo = self._mbconv(i, x, n, h, w, tag)
            if ho < h:
                ends.append(prev)
            h, w = ho, wo
            prev = (x, h, w)
            if trace is not None and i in (0, 5):
                trace[f"mb{i}"] = x.permute(0, 3, 1, 2).clone()
        ends.append(prev)
        (r3, h3, w3), (r4, h4, w4), (r5, h5, w5) = ends[2], ends[3], ends[4]
        u1 = self._up(self.up1, r5, h5, w5, r4, h4, w4, _pad32(112), 2, n, tag + "a")
        if self.up2 is not None:
            f = self._up(self.up2, u1, h4, w4, r3, h3, w3, _pad32(40), 2, n, tag + "b")
        else:       # img_downsample 16: the stride-16 map of up1 is the feature map (lss_submodule.py:152-153)
            f, h3, w3 = u1, h4, w4
        if (h3, w3) != (H // self.ds, W // self.ds):
            raise ValueError(f"camera image {H}x{W}: the stride-{self.ds} feature map is {h3}x{w3}, the frustum expects {H // self.ds}x{W // self.ds}")
        e.conv(self.image_head, f, n, h3, w3, feat)
        if not self.use_gt:
            logit = e.buf(f"cam_logit_{tag}", (n, h3, w3, self.nbins))
            e.conv(self.depth_head, f, n, h3, w3, logit)
            _lib.check(lib.av2x_softmax_channels(P(logit), n * h3 * w3, self.nbins, self.nbins, P(prob), st), "av2x_softmax_channels")
        if trace is not None:
            trace["up1"] = u1.permute(0, 3, 1, 2).clone()
            trace["x_img"] = feat.permute(0, 3, 1, 2).clone()

    def _features_resnet(self, imgs, n, planes, H, W, tag, feat, prob, trace=None):
        """CamEncode_Resnet101.resnet101_forward + heads (lss_submodule.py:262-310) on n images (device NCHW planes)."""
        e, lib, st, P = self.eng, self.lib, self.eng.stream(), lambda t: c_void_p(t.data_ptr())
        x0 = e.buf(f"cam_rin_{tag}", (n, H, W, 16))                       # NHWC, the three colour planes in 16 channel slots (zeros behind them)
        x0.zero_()
        x0[..., :3] = imgs[:, :3].permute(0, 2, 3, 1)
        h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        a = e.buf(f"cam_rstem_{tag}", (n, h, w, 64))
        e.conv(self.r_stem, x0, n, H, W, a)
        hp, wp = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        cur = e.buf(f"cam_rpool_{tag}", (n, hp, wp, 64))
        _lib.check(lib.av2x_maxpool2d(P(a), n, h, w, 64, 3, 2, 1, hp, wp, P(cur), st), "av2x_maxpool2d")
        ch, cw = hp, wp
        for bi, blk in enumerate(self.r_blocks):
            s_ = blk["c2"].stride
            ho, wo = (ch + 2 - 3) // s_ + 1, (cw + 2 - 3) // s_ + 1
            y1 = e.buf(f"cam_rb{bi}a_{tag}", (n, ch, cw, blk["c1"].cout))
            e.conv(blk["c1"], cur, n, ch, cw, y1)
            y2 = e.buf(f"cam_rb{bi}b_{tag}", (n, ho, wo, blk["c2"].cout))
            e.conv(blk["c2"], y1, n, ch, cw, y2)
            idt = cur
            if blk["down"] is not None:
                idt = e.buf(f"cam_rb{bi}d_{tag}", (n, ho, wo, blk["c3"].cout))
                e.conv(blk["down"], cur, n, ch, cw, idt)
            o = e.buf(f"cam_rb{bi}o_{tag}", (n, ho, wo, blk["c3"].cout))
            e.conv(blk["c3"], y2, n, ho, wo, o, residual=idt)
            cur, ch, cw = o, ho, wo
            if trace is not None and bi == 2:
                trace["layer1"] = cur.permute(0, 3, 1, 2).clone()
        if (ch, cw) != (H // self.ds, W // self.ds):
            raise ValueError(f"camera image {H}x{W}: the stride-8 feature map is {ch}x{cw}, the frustum expects {H // self.ds}x{W // self.ds}")
        e.conv(self.image_head, cur, n, ch, cw, feat)
        if not self.use_gt:
            logit = e.buf(f"cam_logit_{tag}", (n, ch, cw, self.nbins))
            e.conv(self.depth_head, cur, n, ch, cw, logit)
            _lib.check(lib.av2x_softmax_channels(P(logit), n * ch * cw, self.nbins, self.nbins, P(prob), st), "av2x_softmax_channels")
        if trace is not None:
            trace["features"] = cur.permute(0, 3, 1, 2).clone()
            trace["x_img"] = feat.permute(0, 3, 1, 2).clone()

    def bev_encode(self, x, n, h, w, out, tag, trace=None):
        """BevEncode.forward (:335-350) on x (n,h,w,C) -> out (n,h,w,outC)."""
        e, lib, st, P = self.eng, self.lib, self.eng.stream(), lambda t: c_void_p(t.data_ptr())
        L = self.bev_stem
        h1, w1 = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
        y = e.buf(f"bev_stem_{tag}", (n, h1, w1, 64))
        e.conv(L, x, n, h, w, y)
        cur, ch, cw = y, h1, w1
        x1 = None
        for bi, blk in enumerate(self.layers):
            s = blk["c1"].stride
            ho, wo = (ch + 2 - 3) // s + 1, (cw + 2 - 3) // s + 1
            c = blk["c1"].cout
            a = e.buf(f"bev_b{bi}a_{tag}", (n, ho, wo, c))
            e.conv(blk["c1"], cur, n, ch, cw, a)
            idt = cur
            if blk["down"] is not None:
                idt = e.buf(f"bev_b{bi}d_{tag}", (n, ho, wo, c))
                e.conv(blk["down"], cur, n, ch, cw, idt)
            o = e.buf(f"bev_b{bi}o_{tag}", (n, ho, wo, c))
            e.conv(blk["c2"], a, n, ho, wo, o, residual=idt)
            cur, ch, cw = o, ho, wo
            if bi == 1:
                x1 = (cur, ch, cw)
                if trace is not None:
                    trace["l1"] = cur.permute(0, 3, 1, 2).clone()
        if trace is not None:
            trace["l3"] = cur.permute(0, 3, 1, 2).clone()
        u = self._up(self.bev_up1, cur, ch, cw, x1[0], x1[1], x1[2], 64, 4, n, tag + "c")
        H2, W2 = x1[1] * 2, x1[2] * 2
        if (H2, W2) != (h, w):
            raise ValueError(f"BevEncode: a {h}x{w} BEV grid does not come back to its own size ({H2}x{W2}); both sides must be multiples of 8")
        big = e.buf(f"bev_big_{tag}", (n, H2, W2, 256))
        _lib.check(lib.av2x_resize_bilinear(P(u), n, x1[1], x1[2], 256, 256, 0, H2, W2, 0, 0, H2, W2, P(big), 256, 0, st), "av2x_resize_bilinear")
        m = e.buf(f"bev_mid_{tag}", (n, H2, W2, 128))
        e.conv(self.bev_up2, big, n, H2, W2, m)
        e.conv(self.bev_out, m, n, H2, W2, out)

    # ------------------------------------------------------------------------------------------------------------ forward
    def forward(self, cam_inputs, out=None, trace=None):
        """batch_merged_cam_inputs of this agent type -> spatial_features (B, ny, nx, outC) NHWC (B = the type's agents).
        ``out``: write there instead of a pool buffer (e.g. the type's rows of the frame's canvas)."""
        e, lib, P = self.eng, self.lib, lambda t: c_void_p(t.data_ptr())
        imgs = cam_inputs["imgs"]
        if imgs.device != e.device or imgs.dtype != torch.float32 or not imgs.is_contiguous():
            imgs = imgs.to(e.device, torch.float32).contiguous()
        B, N, planes, H, W = imgs.shape
        if self.use_gt and planes < 4:
            raise ValueError("use_depth_gt: the images need a 4th (depth) plane")
        if (H // self.ds, W // self.ds) != (self.fH, self.fW):
            raise ValueError(f"camera images are {H}x{W}; data_aug_conf.final_dim says {self.fH * self.ds}x{self.fW * self.ds}")
        params = self._cam_params(cam_inputs)
        ny, nx, tag = int(self.nx[1]), int(self.nx[0]), self.tag
        flat = imgs.view(B * N, planes, H, W)
        feat = e.buf(f"cam_featall_{tag}", (B * N, self.fH, self.fW, self.C))
        prob = None if self.use_gt else e.buf(f"cam_proball_{tag}", (B * N, self.fH, self.fW, self.nbins))
        for i0 in range(0, B * N, self.IMG_CHUNK):
            k = min(self.IMG_CHUNK, B * N - i0)
            self.features(flat[i0:i0 + k], k, planes, H, W, f"{tag}{k}", feat[i0:i0 + k], prob[i0:i0 + k] if prob is not None else None,
                          trace if i0 == 0 else None)
        pooled = e.buf(f"cam_pooled_{tag}", (B, ny, nx, self.C))
        ws = e.buf(f"cam_poolws_{tag}", (int(lib.av2x_lss_pool_workspace_bytes(B, nx, ny, 1, self.C)),), torch.uint8)
        st = e.stream()
        _lib.check(lib.av2x_lss_lift_pool(P(feat), P(prob) if prob is not None else None, P(flat) if self.use_gt else None, planes, H, W, self.ds,
                                          ctypes.cast(self._depth3, c_void_p), self.nbins, self.depth_mode, 0, P(self.frustum), P(params), B, N,
                                          self.fH, self.fW, self.C, ctypes.cast(self._lo, c_void_p), ctypes.cast(self._dx, c_void_p),
                                          ctypes.cast(self._nx, c_void_p), P(ws), P(pooled), st), "av2x_lss_lift_pool")
        if trace is not None:
            trace["pooled"] = pooled.permute(0, 3, 1, 2).clone()
        if out is None:
            out = e.buf(f"cam_bev_{tag}", (B, ny, nx, self.outC))
        for b0 in range(0, B, self.BEV_CHUNK):
            k = min(self.BEV_CHUNK, B - b0)
            self.bev_encode(pooled[b0:b0 + k], k, ny, nx, out[b0:b0 + k], f"{tag}{k}", trace if b0 == 0 else None)
        if trace is not None:
            trace["bev"] = out.permute(0, 3, 1, 2).clone()
        return out
