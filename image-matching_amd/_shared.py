"""Plumbing shared by the drop-in SuperPoint / SuperGlue / Matching classes: one libimx handle
per Matching object (or per standalone model), rebuilt when config or device changes, weights
re-uploaded when a state dict is (re)loaded."""
from collections import OrderedDict

import torch

from . import _lib as L
from .engine import Engine, SG_DEFAULT, SP_DEFAULT


def _snapshot(cfg):
    return repr(sorted((k, repr(v)) for k, v in cfg.items() if k != "weights"))


class Shared:
    def __init__(self):
        self.sp_cfg = None
        self.sg_cfg = None
        self.sp_variant = L.SP_VARIANT_BN
        self.device = None
        self.engine = None
        self._key = None
        self.sd = {L.NET_SUPERPOINT: None, L.NET_SUPERGLUE: None}
        self.dirty = {L.NET_SUPERPOINT: True, L.NET_SUPERGLUE: True}
        self.align_corners = None

    def set_state_dict(self, net, sd):
        self.sd[net] = sd
        self.dirty[net] = True

    def get_engine(self, need):
        dev = self.device if self.device is not None else torch.device("cuda")
        sp = dict(self.sp_cfg) if self.sp_cfg is not None else None
        sg = dict(self.sg_cfg) if self.sg_cfg is not None else None
        if sp is None:      # standalone SuperGlue: descriptor_dim comes from it
            sp = {**SP_DEFAULT, "descriptor_dim": {**SG_DEFAULT, **sg}["descriptor_dim"]}
        if sg is None:
            sg = {**SG_DEFAULT, "descriptor_dim": {**SP_DEFAULT, **sp}["descriptor_dim"]}
        d_sp = {**SP_DEFAULT, **sp}["descriptor_dim"]
        d_sg = {**SG_DEFAULT, **sg}["descriptor_dim"]
        if d_sp != d_sg:
            raise ValueError(f"SuperPoint descriptor_dim {d_sp} != SuperGlue descriptor_dim {d_sg}")
        key = (_snapshot(sp), _snapshot(sg), str(dev), self.sp_variant, self.align_corners)
        if self.engine is None or key != self._key:
            self.engine = Engine(sp, sg, dev, self.sp_variant, self.align_corners)
            self._key = key
            self.dirty = {L.NET_SUPERPOINT: True, L.NET_SUPERGLUE: True}
        for net in need:
            if self.dirty[net]:
                if self.sd[net] is None:
                    raise RuntimeError("no weights loaded for " + ("SuperGlue" if net else "SuperPoint"))
                self.engine.load_state_dict(net, self.sd[net])
                self.dirty[net] = False
        return self.engine


def to_cpu_state_dict(sd):
    out = OrderedDict()
    for k, v in sd.items():
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
        out[k] = t.detach().to("cpu").clone()
    return out


def check_keys(sd, shapes, what):
    """strict load_state_dict semantics: same key set, same shapes (num_batches_tracked optional)."""
    opt = {k for k in shapes if k.endswith("num_batches_tracked")}
    missing = [k for k in shapes if k not in sd and k not in opt]
    unexpected = [k for k in sd if k not in shapes]
    if missing or unexpected:
        msg = f"Error(s) in loading state_dict for {what}:"
        if missing:
            msg += "\n\tMissing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + "."
        if unexpected:
            msg += "\n\tUnexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + "."
        raise RuntimeError(msg)
    for k, shp in shapes.items():
        if k in sd and tuple(sd[k].shape) != tuple(shp) and not (sd[k].numel() == 1 and len(shp) == 0):
            raise RuntimeError(f"Error(s) in loading state_dict for {what}:\n\tsize mismatch for {k}: "
                               f"copying a param with shape {tuple(sd[k].shape)} from checkpoint, "
                               f"the shape in current model is {tuple(shp)}.")


class ModelBase(torch.nn.Module):
    """nn.Module facade: .eval()/.to()/.cuda()/.state_dict()/.load_state_dict() behave as callers of
    the reference expect; parameters live in the libimx handle, not in torch."""
    _net = None

    def _init_shared(self, shared):
        object.__setattr__(self, "_shared", shared if shared is not None else Shared())

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if isinstance(dev, (str, torch.device, int)):
            dev = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
            self._shared.device = dev
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device) if isinstance(device, int) else (device or "cuda"))

    def cpu(self):
        raise RuntimeError("image_matching_amd models run on the GPU only (no CPU path)")

    def train(self, mode=True):
        if mode:
            raise RuntimeError("image_matching_amd is inference-only; call .eval()")
        return super().train(False)

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._shared.sd[self._net].items())
