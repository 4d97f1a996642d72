"""CPU tests of the host-side logic: state_dict compatibility, host positive counts, synthetic data generator."""
import io

import numpy as np
import torch

import densebox_amd as D
from densebox_amd import synth, labels as LB
from oracle import densebox_oracle as O


def test_state_dict_round_trip_with_aliased_keys(golden):
    """A checkpoint in the reference's format (aliased keys, DenseBox.py:2206) loads with strict=True."""
    for kind in ('DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC'):
        a = getattr(D, kind)(synth.vgg19_standin(seed=0))
        synth.fill_params_(a, 5)
        keys = [str(k) for k in golden('net_' + kind)['keys']]
        sd = a.state_dict()
        assert list(sd.keys()) == keys
        buf = io.BytesIO()
        torch.save(sd, buf)
        buf.seek(0)
        b = getattr(D, kind)(synth.vgg19_standin(seed=1))
        b.load_state_dict(torch.load(buf), strict=True)
        for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
            assert n1 == n2 and torch.equal(p1, p2)
        # conv3_3 exists in the checkpoint although forward never runs it (DenseBox.py:97-102, :193-195)
        assert 'conv3_3_1.weight' in sd and 'conv3_3.0.weight' in sd
        # aliased registrations share storage
        assert sd['conv1_1_1.weight'].data_ptr() == sd['conv1_1.0.weight'].data_ptr()
        assert sd['conv5_1_det.weight'].data_ptr() == sd['output_score.0.weight'].data_ptr()


def test_param_counts():
    """11 375 173 / 11 876 426 / 12 274 258 unique parameters incl. the unused conv3_3."""
    want = {'DenseBox': 11375173, 'DenseBoxLM': 11876426, 'DenseBoxLMLOC': 12274258}
    for kind, n in want.items():
        net = getattr(D, kind)(synth.vgg19_standin(seed=0))
        assert sum(p.numel() for p in net.parameters()) == n


def test_positive_count_host_matches_oracle_maps(golden):
    g = golden('labels')
    for labels in (None, g['lab']):
        want = O.init_score_map(g['bbox'], labels).sum(axis=(1, 2, 3)).astype(np.int64)
        assert np.array_equal(LB.positive_count(g['bbox'], labels), want)
    bbox, _, lab = synth.synth_labels(256, seed=9, neg_frac=0.2)
    want = O.init_score_map(bbox.numpy(), lab.numpy()).sum(axis=(1, 2, 3)).astype(np.int64)
    assert np.array_equal(LB.positive_count(bbox, lab), want)
    assert LB.neg_counts(42, 2) == O.neg_counts(42, 2) == (21, 11)


def test_synth_is_deterministic_and_in_range():
    x1, b1, v1, l1 = synth.synth_batch(16, seed=3)
    x2, b2, v2, l2 = synth.synth_batch(16, seed=3)
    assert torch.equal(x1, x2) and torch.equal(b1, b2) and torch.equal(v1, v2) and torch.equal(l1, l2)
    pos = l1[:, 0] == 1
    assert (b1[pos] >= 2.0).all() and (b1[pos] <= 58.0).all()
    assert (v1[pos] >= 2.0).all() and (v1[pos] <= 58.0).all()          # landmarks >= 8 px inside the patch
    assert (b1[~pos] == 0).all() and (v1[~pos] == 0).all()
    assert tuple(x1.shape) == (16, 3, 240, 240) and abs(float(x1.mean())) < 0.5
