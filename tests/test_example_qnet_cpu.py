"""CPU: the GEMM formulation of the example Q-net (examples/config3_dqn_inference.py) equals its
literal nn.Conv2d evaluation (architecture of net.py:137-150 / forward net.py:81-102).  Floating
point: fp32, tolerance 1e-5 relative to the output scale."""
import importlib.util
import os

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_forward_equals_conv_forward():
    spec = importlib.util.spec_from_file_location("cfg3", os.path.join(REPO, "examples", "config3_dqn_inference.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    net = mod.QNetSimplify().eval()
    x = (torch.rand(37, 7, 15, 4) < 0.3).float() * torch.rand(37, 7, 15, 4)
    with torch.no_grad():
        a, b = net.forward_conv(x), net(x)
    assert a.shape == b.shape == (37, 1)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max()))
    keys = set(net.state_dict())
    assert {"conv1.weight", "conv4.bias", "conv_shunzi.weight", "fc1.weight", "fc2.bias"} <= keys
    assert net.state_dict()["fc1.weight"].shape == (256, 4864)   # net.py:147
