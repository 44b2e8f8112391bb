"""Encoder interface parity (CPU): same state-dict keys/shapes as the reference's VOSNet and the same outputs
for the same (deterministically generated) weights.  Goldens come from the reference's own modules."""
import importlib
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import inputs as gin

GOLD = Path(__file__).resolve().parent / 'golden'


@pytest.fixture(scope='module')
def vos_net():
    return importlib.import_module('semi-supervised-vos_amd.vos_net')


@pytest.mark.parametrize('name', ['resnet18', 'resnet50', 'resnet101'])
def test_state_dict_keys_match_reference(vos_net, name):
    want = json.loads((GOLD / 'encoder_keys.json').read_text())[name]
    sd = vos_net.VOSNet(name).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == want


@pytest.mark.parametrize('name', ['resnet18', 'resnet50'])
def test_forward_matches_reference(vos_net, goldens, name):
    net = vos_net.VOSNet(name)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gin.fill_state_dict(net.state_dict()).items()})
    net.eval()
    with torch.no_grad():
        y = net(torch.from_numpy(gin.encoder_input())).numpy()
    g = goldens[f'enc_{name}_out']
    assert y.shape == g.shape == (1, 256, 8, 12)
    # conv algorithm choice depends on the thread count: compare relative to the output scale
    assert np.abs(y - g).max() <= 1e-4 * np.abs(g).max(), (np.abs(y - g).max(), np.abs(g).max())


def test_facebook_is_a_clear_error(vos_net):
    with pytest.raises(NotImplementedError):
        vos_net.VOSNet('facebook')


@pytest.mark.parametrize('name', ['resnet18', 'resnet50'])
def test_folded_batchnorm_is_equivalent(vos_net, name):
    net = vos_net.VOSNet(name)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gin.fill_state_dict(net.state_dict()).items()})
    net.eval()
    x = torch.from_numpy(gin.encoder_input())
    with torch.no_grad():
        y0 = net(x)
        y1 = net.fold_batchnorm()(x)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
    assert (y0 - y1).abs().max() <= 1e-4 * y0.abs().max()


@pytest.mark.parametrize('model', ['resnet18', 'resnet50'])
def test_fused_epilogue_forward_equals_module_forward(model):
    """prepare_for_inference(fuse_epilogue=True): conv without bias + bias_act_ (torch fallback on the CPU) is the same
    function as the folded module chain."""
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    torch.manual_seed(1)
    a = vn.VOSNet(model)
    b = vn.VOSNet(model)
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 3, 40, 56)
    a.prepare_for_inference(torch.device('cpu'), None, fuse_epilogue=False)
    b.prepare_for_inference(torch.device('cpu'), None, fuse_epilogue=True)
    assert b.fused and not a.fused
    with torch.no_grad():
        ya, yb = a(x), b(x)
    assert ya.shape == yb.shape == (2, 256, 5, 7)
    assert float((ya - yb).abs().max()) <= 1e-5 * float(ya.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('relu,res', [(True, True), (True, False), (False, True), (False, False)])
def test_bias_act_kernel(dtype, relu, res):
    """vosprop_bias_act against the torch ops it replaces (f32 arithmetic, one rounding)."""
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    torch.manual_seed(0)
    dev = torch.device('cuda', 0)
    for (n, c, h, w) in [(3, 64, 17, 23), (1, 8, 1, 5), (16, 256, 60, 107)]:
        y0 = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        b = torch.randn(c, device=dev).to(dtype)
        r = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last) if res else None
        want = y0.float() + b.float().view(1, -1, 1, 1) + (r.float() if res else 0.0)
        want = (want.relu() if relu else want).to(dtype)
        y = y0.clone(memory_format=torch.preserve_format)
        out = vn.bias_act_(y, b, r, relu)
        assert out.data_ptr() == y.data_ptr()
        assert torch.equal(y, want), float((y.float() - want.float()).abs().max())


@pytest.mark.gpu
def test_fused_encoder_on_gpu_matches_f32_reference():
    """The encoder as the engine runs it (bf16, channels_last, folded BN, fused epilogues, HIP-graph replay) against the same
    network in f32 on the CPU: relative error of a bf16 ResNet-50 forward."""
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    torch.manual_seed(2)
    ref = vn.VOSNet('resnet50').eval()
    net = vn.VOSNet('resnet50')
    net.load_state_dict(ref.state_dict())
    x = torch.randn(4, 3, 96, 160)
    with torch.no_grad():
        want = ref(x)
    dev = torch.device('cuda', 0)
    net.prepare_for_inference(dev, torch.bfloat16)
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        eager = net(xg).float().cpu()
        g = vn.GraphedEncoder(net)
        g(xg)
        graphed = g(xg).float().cpu()
    scale = float(want.abs().max())
    assert float((eager - want).abs().max()) <= 0.05 * scale
    assert g.graph is not None and not g.failed
    # MIOpen may pick another algorithm under stream capture: same function, bf16-level differences
    assert float((graphed - eager).abs().max()) <= 0.02 * scale
    assert float((graphed - want).abs().max()) <= 0.05 * scale
