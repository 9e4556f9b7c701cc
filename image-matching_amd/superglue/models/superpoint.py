"""Drop-in for the official (Magic Leap) SuperPoint used by the reference's
superglue/models/superpoint.py:95-202: no BatchNorm, input is {'image': x}, dense descriptors
L2-normalised with F.normalize.  Same kernels as the BN variant, un-folded weights."""
from pathlib import Path

import torch

from ... import _lib as L
from ... import synth
from ..._shared import ModelBase, check_keys, to_cpu_state_dict


class SuperPoint(ModelBase):
    default_config = {
        'descriptor_dim': 256,
        'nms_radius': 4,
        'keypoint_threshold': 0.005,
        'max_keypoints': -1,
        'remove_borders': 4,
    }
    _net = L.NET_SUPERPOINT
    _variant = L.SP_VARIANT_OFFICIAL

    def __init__(self, config, _shared=None):
        super().__init__()
        self._init_shared(_shared)
        self.config = {**self.default_config, **config}
        self._shared.sp_cfg = self.config
        self._shared.sp_variant = self._variant
        self._shapes = synth.superpoint_official_shapes(self.config['descriptor_dim'])
        self._shared.set_state_dict(self._net, to_cpu_state_dict(synth.synth_state_dict(self._shapes, 0)))
        # the reference always loads weights/superpoint_v1.pth next to the module (:136-137);
        # 'weights_path': None skips it (parity tests load a state dict afterwards)
        path = self.config.get('weights_path', Path(__file__).parent / 'weights/superpoint_v1.pth')
        if path is not None:
            self.load_state_dict(torch.load(str(path), map_location='cpu'))
        mk = self.config['max_keypoints']
        if mk == 0 or mk < -1:
            raise ValueError('\"max_keypoints\" must be positive or \"-1\"')
        if path is not None:
            print('Loaded SuperPoint model')
        self.train(False)

    def load_state_dict(self, state_dict, strict=True):
        sd = to_cpu_state_dict(state_dict)
        check_keys(sd, self._shapes, type(self).__name__)
        self._shared.set_state_dict(self._net, sd)

    def forward(self, data):
        eng = self._shared.get_engine([self._net])
        kpts, scores, desc, n = eng.superpoint(data['image'])
        return {
            'keypoints': [kpts[b, :n[b]] for b in range(len(n))],
            'scores': tuple(scores[b, :n[b]] for b in range(len(n))),
            'descriptors': [desc[b, :n[b]].t() for b in range(len(n))],
        }
