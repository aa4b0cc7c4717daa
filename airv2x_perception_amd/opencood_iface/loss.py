"""Mirror of the reference's ``opencood/loss/point_pillar_loss_multiclass.py`` (PointPillarLossMultiClass :79-179): same
constructor argument (``{"cls_weight", "reg", "num_class"}``), ``forward(output_dict, target_dict, prefix="")`` returning the
total loss, ``loss_dict`` with the python floats the reference logs, ``logging(epoch, batch_id, batch_len, writer)``.

The forward AND the gradient with respect to ``psm`` / ``rm`` / ``obj`` are one fused pass over the anchors in
libairv2x_hip.so (``av2x_pp_loss``); the returned tensor carries a ``torch.autograd.Function`` whose backward hands those
gradients (scaled by the incoming one) to whatever produced the head maps.  GPU only."""
from __future__ import annotations

from ctypes import c_void_p

import torch
import torch.nn as nn

from .. import _lib


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class _LossDict(dict):
    """``loss_dict`` of the reference holds python floats read back with three ``.item()`` calls inside ``forward``
    (point_pillar_loss_multiclass.py:172-177) -- a host wait for the whole queued forward before the backward can be issued.  Here the
    values (and the class-id range check) travel to a pinned host buffer asynchronously and become floats at the FIRST READ of the
    dict (``logging``, ``loss_dict[...]``, iteration) -- or at a later ``forward`` once the copy has landed; an out-of-range class id
    raises there."""

    def __init__(self):
        super().__init__()
        self._pending = []

    def _push(self, prefix, host, event, num_class, checked):
        self._pending.append((prefix, host, event, num_class, checked))

    def _flush(self, block=True):
        """block=False (the next forward): only the read-backs that have already landed, in order -- no host wait."""
        while self._pending:
            prefix, host, event, num_class, checked = self._pending[0]
            if not block and not event.query():
                return
            self._pending.pop(0)
            event.synchronize()
            vals = host.tolist()
            dict.update(self, {"total_loss" + prefix: vals[0], "reg_loss" + prefix: vals[1], "conf_loss" + prefix: vals[2]})
            if checked:
                lo, hi = int(vals[-2]), int(vals[-1])
                if lo < 0 or hi >= num_class:
                    raise IndexError(f"class_ids outside [0, {num_class}): min {lo}, max {hi}")

    def __getitem__(self, k):
        self._flush()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._flush()
        return dict.get(self, k, default)

    def items(self):
        self._flush()
        return dict.items(self)

    def values(self):
        self._flush()
        return dict.values(self)

    def keys(self):
        self._flush()
        return dict.keys(self)

    def __iter__(self):
        self._flush()
        return dict.__iter__(self)

    def __len__(self):
        self._flush()
        return dict.__len__(self)

    def __contains__(self, k):
        self._flush()
        return dict.__contains__(self, k)

    def __repr__(self):
        self._flush()
        return dict.__repr__(self)

    def copy(self):
        self._flush()
        return dict(self)

    def __reduce__(self):                 # pickling / deepcopy of the criterion: the floats, as a plain dict (events do not travel)
        self._flush()
        return (dict, (dict(dict.items(self)),))


class _PPLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, psm, rm, obj, targets, pos, cls, num_class, cls_weight, reg_coe):
        lib = _lib.load()
        B, AC, H, W = psm.shape
        A = AC // num_class
        if rm.shape != (B, A * 7, H, W) or obj.shape != (B, A, H, W):
            raise ValueError(f"head shapes psm {tuple(psm.shape)} rm {tuple(rm.shape)} obj {tuple(obj.shape)} are inconsistent")
        if targets.numel() != B * H * W * A * 7 or pos.numel() != B * H * W * A or cls.numel() != B * H * W * A:
            raise ValueError("label tensors do not match the head maps")
        dev = psm.device
        need = any(ctx.needs_input_grad[:3])
        ws = torch.empty(lib.av2x_pp_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(t, memory_format=torch.contiguous_format) if need else None for t in (psm, rm, obj)]
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.av2x_pp_loss(_p(psm), _p(rm), _p(obj), _p(targets), _p(pos), _p(cls), B, H, W, A, num_class,
                                    float(cls_weight), float(reg_coe), _p(ws), _p(out4), _p(grads[0]), _p(grads[1]), _p(grads[2]),
                                    st), "av2x_pp_loss")
        ctx.grads = grads
        ctx.mark_non_differentiable(out4)
        return out4[0].clone(), out4

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        g = ctx.grads
        out = [None] * 9
        for i in range(3):
            if ctx.needs_input_grad[i]:
                out[i] = g[i] * g_total
        return tuple(out)


class PointPillarLossMultiClass(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.alpha, self.gamma = 0.25, 2.0          # fixed in the reference (:83-84); compiled into the kernel
        self.cls_weight = args["cls_weight"]
        self.reg_coe = args["reg"]
        self.flow_weight = args["flow_weight"] if "flow_weight" in args else 1.0
        self.loss_dict = _LossDict()
        self.use_dir = False
        self.cls_num = args["num_class"]
        # class-id range check (the reference's one_hot scatter_ raises on an out-of-range id, point_pillar_loss_multiclass.py:118-125):
        #   True    (default) checked asynchronously -- the verdict rides with the lazy loss read-back, so a bad id raises IndexError at the
        #           first READ of loss_dict (logging(), repr, the next forward's flush), i.e. AFTER this step's backward / optimizer.step();
        #   "sync"  checked before forward() returns (one blocking 8-byte read-back per step): raises where the reference raises;
        #   False   not checked (the kernel trains such an anchor as background).
        self.validate_class_ids = True

    def forward(self, output_dict, target_dict, prefix=""):
        psm, rm, obj = (output_dict[k + prefix] for k in ("psm", "rm", "obj"))
        if psm.device.type != "cuda":
            raise RuntimeError("PointPillarLossMultiClass (MI355X build) has no CPU path")
        if psm.device.index is not None and psm.device.index != torch.cuda.current_device():
            raise RuntimeError(f"the loss kernel launches on the current HIP device (cuda:{torch.cuda.current_device()}), psm is on {psm.device}")
        ac = psm.shape[1] if psm.dim() == 4 else psm.shape[-1]
        if ac % int(self.cls_num):
            raise ValueError(f"psm has {ac} channels, not a multiple of num_class = {self.cls_num} (loss args vs model anchor_number * num_class)")
        cid = target_dict["class_ids"]
        if cid.numel() and self.validate_class_ids:
            # the reference's one_hot scatter_ raises on an out-of-range class id (point_pillar_loss_multiclass.py:118-125); the kernel
            # would silently train such an anchor as background.  One small reduction + read-back per step (set
            # validate_class_ids = False to skip it once the label pipeline is trusted)
            # ... asynchronously: the two reductions are queued with the step and read together with the loss parts below, so the
            # check no longer drains the queued forward before the loss and the backward can be enqueued
            cid_range = torch.stack([cid.min(), cid.max()]).to(torch.float32)
            if self.validate_class_ids == "sync":
                lo, hi = (int(v) for v in cid_range.tolist())
                if lo < 0 or hi >= int(self.cls_num):
                    raise IndexError(f"class_ids out of range [0, {int(self.cls_num)}): min {lo}, max {hi} (the reference's one_hot scatter_ raises here)")
        f32 = lambda t: t.detach().to(psm.device, torch.float32).contiguous()
        cont = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        total, parts = _PPLoss.apply(cont(psm), cont(rm), cont(obj), f32(target_dict["targets"]), f32(target_dict["pos_equal_one"]),
                                     target_dict["class_ids"].detach().to(psm.device, torch.int32).contiguous(),
                                     int(self.cls_num), self.cls_weight, self.reg_coe)
        if not isinstance(self.loss_dict, _LossDict):      # a caller replaced it with a plain dict: keep its entries, restore the lazy one
            ld = _LossDict()
            dict.update(ld, self.loss_dict)
            self.loss_dict = ld
        self.loss_dict._flush(block=False)                 # read-backs of earlier calls that have landed (their class-id verdict with them)
        checked = bool(cid.numel() and self.validate_class_ids and self.validate_class_ids != "sync")
        dev_vals = torch.cat([parts.detach().float().flatten(), cid_range.to(parts.device)]) if checked else parts.detach().float().flatten()
        host = torch.empty(dev_vals.numel(), dtype=torch.float32).pin_memory()
        host.copy_(dev_vals, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(psm.device))     # the stream the copy above was queued on (psm's device, checked above)
        self.loss_dict._push(prefix, host, event, int(self.cls_num), checked)
        return total

    def logging(self, epoch, batch_id, batch_len, writer=None):
        """:295-330: the progress line, and one scalar per entry of ``loss_dict`` on the tensorboard writer."""
        total = [v for k, v in self.loss_dict.items() if "total_loss" in k]
        total = sum(total) if len(total) > 1 else total[0]
        msg = "[epoch {}][{}/{}], || Loss: {:.2f} ||".format(epoch, batch_id + 1, batch_len, total)
        for k, v in self.loss_dict.items():
            msg += "{}: {:.2f} | ".format(k.replace("_loss", "").replace("_single", ""), v)
        if writer is not None:
            for k, v in self.loss_dict.items():
                writer.add_scalar(k, v, epoch * batch_len + batch_id)
        return msg
