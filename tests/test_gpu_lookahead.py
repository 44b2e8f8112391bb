"""Encoder look-ahead (inference_utils.encoded_branches) on the GPU: batches run across video boundaries and a short batch is
padded to the full size, so the encoder is only ever called with ONE batch shape per frame size - and every frame still gets its
own features, in loader order, under its own video name."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


class ShapeRecordingEncoder:
    def __init__(self):
        self.shapes = []

    def __call__(self, x):
        self.shapes.append(tuple(x.shape))
        return x[:, :1, ::8, ::8].float() * 2.0          # per-sample, no mixing across the batch - like the real encoder


def test_one_encoder_shape_across_videos_and_tails():
    iu = importlib.import_module('semi-supervised-vos_amd.inference_utils')
    dev = torch.device('cuda', 0)
    lengths = {'a': 5, 'b': 3, 'c': 7}
    frames, k = [], 0
    for name, n in lengths.items():
        for _ in range(n):
            frames.append((torch.full((1, 3, 16, 24), float(k)), (name,)))
            k += 1
    enc = ShapeRecordingEncoder()
    got = list(iu.encoded_branches([enc], frames, dev, None, batch=4))
    assert set(enc.shapes) == {(4, 3, 16, 24)} and len(enc.shapes) == 4       # 15 frames -> 4 calls, the last one padded
    assert [name for _, name in got] == [n for n, c in lengths.items() for _ in range(c)]
    for i, (feats, _) in enumerate(got):
        assert feats[0].shape == (1, 1, 2, 3)
        assert torch.all(feats[0] == 2.0 * i)

    # a change of frame size flushes (and pads) the pending batch; batch 1 never pads
    mixed = [(torch.zeros(1, 3, 16, 24), ('a',))] * 2 + [(torch.zeros(1, 3, 8, 8), ('b',))] * 3
    enc = ShapeRecordingEncoder()
    assert len(list(iu.encoded_branches([enc], mixed, dev, None, batch=4))) == 5
    assert enc.shapes == [(4, 3, 16, 24), (4, 3, 8, 8)]
    enc = ShapeRecordingEncoder()
    assert len(list(iu.encoded_branches([enc], mixed, dev, None, batch=1))) == 5
    assert {s[0] for s in enc.shapes} == {1}
