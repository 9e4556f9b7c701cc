"""Drop-in for the reference's dense SuperPoint forward (superpoint/models/superpoint_train.py:8-57),
the network used by training and by pseudo-label export (superpoint_export_pseudo.py:78): same
constructor (`descriptor_length`), `forward(x) -> {'semi', 'desc'}` and `self.output`; inference
only (the encoder, heads and channel normalisation run in libimx)."""
from ... import _lib as L
from ... import synth
from ..._shared import ModelBase, check_keys, to_cpu_state_dict


class SuperPoint(ModelBase):
    """ SuperPoint network, dense outputs: semi (N,65,H/8,W/8), desc (N,d,H/8,W/8). """
    _net = L.NET_SUPERPOINT

    def __init__(self, descriptor_length=256, _shared=None):
        super().__init__()
        self._init_shared(_shared)
        self.config = {'descriptor_dim': descriptor_length, 'nms_radius': 4, 'keypoint_threshold': 0.005,
                       'max_keypoints': -1, 'remove_borders': 4}
        self._shared.sp_cfg = self.config
        self._shapes = synth.superpoint_bn_shapes(descriptor_length)
        self._shared.set_state_dict(self._net, to_cpu_state_dict(synth.synth_state_dict(self._shapes, 0)))
        self.output = None
        self.train(False)

    def load_state_dict(self, state_dict, strict=True):
        sd = to_cpu_state_dict(state_dict)
        check_keys(sd, self._shapes, type(self).__name__)
        self._shared.set_state_dict(self._net, sd)

    def forward(self, x):
        eng = self._shared.get_engine([self._net])
        semi, desc = eng.superpoint_dense(x)
        output = {'semi': semi, 'desc': desc}
        self.output = output
        return output
