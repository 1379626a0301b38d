"""The NN restatement (c4a0_amd/nn.py) against golden vectors produced by the REFERENCE's own
src/c4a0/nn.py (tests/golden/make_nn_fixture.py, run in the development container).
Floating point: tolerance 1e-5 absolute in f32 (different op order after BN folding)."""
import os

import numpy as np
import pytest
import torch

from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig, flops_per_leaf

FIX = os.path.join(os.path.dirname(__file__), "golden", "nn_fixture.npz")
TOL = 1e-5


def _load():
    z = np.load(FIX)
    cfg = ModelConfig(*[int(v) for v in z["cfg"]])
    model = ConnectFourNet(cfg)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return z, cfg, model.eval()


def test_state_dict_keys_are_the_references():
    z, cfg, model = _load()
    assert set(model.state_dict().keys()) == {k[3:] for k in z.files if k.startswith("sd/")}
    assert "conv.1.block.2.running_mean" in model.state_dict() and "fc_policy.0.0.weight" in model.state_dict()


def test_module_forward_matches_reference_outputs():
    z, cfg, model = _load()
    with torch.no_grad():
        lp, qp, qn = model(torch.from_numpy(z["x"]))
    assert np.abs(lp.numpy() - z["policy_logprobs"]).max() <= TOL
    assert np.abs(qp.numpy() - z["q_penalty"]).max() <= TOL
    assert np.abs(qn.numpy() - z["q_no_penalty"]).max() <= TOL


def test_bn_folded_f32_inference_matches_reference_outputs():
    z, cfg, model = _load()
    net = InferenceNet(model, torch.device("cpu"), dtype=torch.float32)
    lp, q = net(torch.from_numpy(z["x"]))
    assert lp.dtype == torch.float32 and q.shape == (z["x"].shape[0], 2)
    assert np.abs(lp.numpy() - z["policy_logprobs"]).max() <= TOL
    assert np.abs(q[:, 0].numpy() - z["q_penalty"]).max() <= TOL
    assert np.abs(q[:, 1].numpy() - z["q_no_penalty"]).max() <= TOL
    assert np.allclose(np.exp(lp.numpy()).sum(1), 1.0, atol=1e-5)   # nn_test.py:26-40: exp(logprobs) sums to 1
    assert np.all(np.abs(q.numpy()) <= 1.0)
    out_lp, out_q = torch.empty(24, 7), torch.empty(24, 2)
    net(torch.from_numpy(z["x"]), out_logprobs=out_lp, out_q=out_q)
    assert torch.equal(out_lp, lp) and torch.equal(out_q, q)


def test_restatement_matches_the_reference_at_the_hip_kernels_width():
    """tests/golden/nn_fixture_1x32.npz: outputs of the reference's nn.py for a 1-block / 32-channel net with 2 policy /
    2 value layers whose weights are a closed-form function of (tensor name, index) -- the width the HIP kernels
    accept, so tests/test_gpu_nn.py can compare them with the reference itself.  Here: the f32 restatement, 1e-5."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from closed_form_weights import fill_closed_form

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nn_fixture_1x32.npz"))
    assert [int(v) for v in z["cfg"]] == [1, 32, 2, 2]
    model = ConnectFourNet(ModelConfig(*[int(v) for v in z["cfg"]])).eval()
    with torch.no_grad():
        fill_closed_form(model)
        lp, qp, qn = model(torch.from_numpy(z["x"]))
    assert np.abs(lp.numpy() - z["policy_logprobs"]).max() <= TOL
    assert np.abs(qp.numpy() - z["q_penalty"]).max() <= TOL and np.abs(qn.numpy() - z["q_no_penalty"]).max() <= TOL
    net = InferenceNet(model, torch.device("cpu"), dtype=torch.float32)   # BN folded
    lp2, q2 = net(torch.from_numpy(z["x"]))
    assert np.abs(lp2.numpy() - z["policy_logprobs"]).max() <= TOL and np.abs(q2[:, 0].numpy() - z["q_penalty"]).max() <= TOL
    # the outputs are informative (not saturated, not uniform): the comparison on the GPU means something
    assert 0.1 < z["policy_logprobs"].std(0).min() and np.abs(z["q_penalty"]).max() < 0.999


def test_forward_numpy_on_the_cpu_path_matches_reference_outputs():
    """InferenceNet.forward_numpy (reference nn.py:119-130) without a HIP device: the plain PyTorch path, same contract."""
    z, cfg, model = _load()
    net = InferenceNet(model, torch.device("cpu"), dtype=torch.float32)
    lp, qp, qn = net.forward_numpy(z["x"])
    assert lp.dtype == np.float32 and lp.shape == (24, 7) and qp.shape == (24,) and lp.flags["C_CONTIGUOUS"] and qn.flags["C_CONTIGUOUS"]
    assert np.abs(lp - z["policy_logprobs"]).max() <= TOL and np.abs(qp - z["q_penalty"]).max() <= TOL and np.abs(qn - z["q_no_penalty"]).max() <= TOL


def test_flops_per_leaf_matches_survey():
    # 1x32 16.1 M; 4x32 20.7 M; 8x64 107.5 M
    assert round(flops_per_leaf(ModelConfig(1, 32, 4, 2)) / 1e6, 1) == 16.1
    assert round(flops_per_leaf(ModelConfig(4, 32, 4, 2)) / 1e6, 1) == 20.7
    assert round(flops_per_leaf(ModelConfig(8, 64, 4, 2)) / 1e6, 1) == 107.5


def test_param_counts_match_survey():
    # SURVEY 8a a20 [probe]: 1x32 7 272 745; 4x32 7 328 425; 8x64 29 550 921
    n = lambda cfg: sum(p.numel() for p in ConnectFourNet(cfg).parameters())
    assert n(ModelConfig(1, 32, 4, 2)) == 7_272_745
    assert n(ModelConfig(4, 32, 4, 2)) == 7_328_425
    assert n(ModelConfig(8, 64, 4, 2)) == 29_550_921
