"""Drop-in for the reference's inference SuperGlue (superglue/models/superglue_test.py:177-285):
same config keys, `forward(data)` contract and state-dict key names; the keypoint encoder,
attentional GNN, optimal transport and match extraction run in libimx HIP kernels."""
import torch

from ... import _lib as L
from ... import synth
from ..._shared import ModelBase, check_keys, to_cpu_state_dict


class SuperGlue(ModelBase):
    default_config = {
        'descriptor_dim': 256,
        'weights': 'indoor',
        'keypoint_encoder': [32, 64, 128, 256],
        'GNN_layers': ['self', 'cross'] * 9,
        'sinkhorn_iterations': 100,
        'match_threshold': 0.2,
    }
    _net = L.NET_SUPERGLUE

    def __init__(self, config, _shared=None):
        super().__init__()
        self._init_shared(_shared)
        self.config = {**self.default_config, **config}
        self._shared.sg_cfg = self.config
        self._shapes = synth.superglue_shapes(self.config['descriptor_dim'], self.config['keypoint_encoder'],
                                              len(self.config['GNN_layers']))
        self._shared.set_state_dict(self._net, to_cpu_state_dict(synth.synth_state_dict(self._shapes, 0)))
        if self.config['weights']:
            checkpoints = torch.load(config['weights'], map_location='cpu')    # (:222) indexes the user's config
            if 'indoor' in self.config['weights'] or 'outdoor' in self.config['weights']:
                state_dict = checkpoints
            else:
                state_dict = checkpoints['net']
            self.load_state_dict(state_dict)
            print('Loaded SuperGlue model weights')
        self.train(False)

    def load_state_dict(self, state_dict, strict=True):
        sd = to_cpu_state_dict(state_dict)
        check_keys(sd, self._shapes, type(self).__name__)
        self._shared.set_state_dict(self._net, sd)

    def forward(self, data):
        """Run SuperGlue on a pair of keypoint sets: descriptors{0,1} (B,d,N), keypoints{0,1}
        (B,N,2), scores{0,1} (B,N), image{0,1} (only .shape is used)."""
        desc0, desc1 = data['descriptors0'], data['descriptors1']
        kpts0, kpts1 = data['keypoints0'], data['keypoints1']
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:     # no keypoints (:235-242): int32 matches
            shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {
                'matches0': kpts0.new_full(shape0, -1, dtype=torch.int),
                'matches1': kpts1.new_full(shape1, -1, dtype=torch.int),
                'matching_scores0': kpts0.new_zeros(shape0),
                'matching_scores1': kpts1.new_zeros(shape1),
            }
        eng = self._shared.get_engine([self._net])
        m0, m1, ms0, ms1 = eng.superglue(kpts0, data['scores0'], desc0, data['image0'].shape,
                                         kpts1, data['scores1'], desc1, data['image1'].shape)
        return {
            'matches0': m0,  # use -1 for invalid match
            'matches1': m1,  # use -1 for invalid match
            'matching_scores0': ms0,
            'matching_scores1': ms1,
        }
