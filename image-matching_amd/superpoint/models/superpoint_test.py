"""Drop-in for the reference's inference SuperPoint (BatchNorm variant):
superpoint/models/superpoint_test.py:55-161 — same constructor config, `forward(x)` signature,
returned dict, state-dict key names and `.config` attribute; the arithmetic runs in libimx
(HIP kernels for gfx950) instead of torch ops."""
import torch

from ... import _lib as L
from ... import synth
from ..._shared import ModelBase, check_keys, to_cpu_state_dict


class SuperPoint(ModelBase):
    """SuperPoint detector/descriptor, MI355X-native.  Input (B,1,H,W) float32 in [0,1] on the GPU;
    output {'keypoints': list[(K,2) (x,y)], 'scores': tuple[(K,)], 'descriptors': list[(d,K)]}."""
    default_config = {
        'descriptor_dim': 256,
        'nms_radius': 4,
        'keypoint_threshold': 0.005,
        'max_keypoints': -1,
        'remove_borders': 4,
    }
    _net = L.NET_SUPERPOINT
    _variant = L.SP_VARIANT_BN

    def __init__(self, config, _shared=None):
        super().__init__()
        self._init_shared(_shared)
        self.config = {**self.default_config, **config}
        self._shared.sp_cfg = self.config
        self._shared.sp_variant = self._variant
        self._shapes = synth.superpoint_bn_shapes(self.config['descriptor_dim'])
        # no checkpoint: deterministic synthetic parameters (the reference would hold torch's random init)
        self._shared.set_state_dict(self._net, to_cpu_state_dict(synth.synth_state_dict(self._shapes, 0)))
        if self.config['weights']:      # KeyError when the key is absent, as in the reference (:87)
            checkpoints = torch.load(self.config['weights'], map_location='cpu')
            pretrained_dict = checkpoints['model_state_dict']
            new_state_dict = {}
            for k, v in pretrained_dict.items():    # multi-GPU checkpoints carry a 'module.' prefix (:93-96)
                new_state_dict[k[7:] if "module" in k else k] = v
            self.load_state_dict(new_state_dict)
            print("Loaded SuperPoint model")
        self.train(False)

    def load_state_dict(self, state_dict, strict=True):
        sd = to_cpu_state_dict(state_dict)
        check_keys(sd, self._shapes, type(self).__name__)
        self._shared.set_state_dict(self._net, sd)

    def forward(self, x):
        eng = self._shared.get_engine([self._net])
        kpts, scores, desc, n = eng.superpoint(x)
        return {
            'keypoints': [kpts[b, :n[b]] for b in range(len(n))],
            'scores': tuple(scores[b, :n[b]] for b in range(len(n))),
            'descriptors': [desc[b, :n[b]].t() for b in range(len(n))],     # (d, K) view
        }
