"""GPU checks of the evaluator: the hand-written MFMA conv tower and the bf16 inference net
against a plain PyTorch fp32 reference of the same op (floating point => tolerance, stated)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _positions(n, seed=3):
    from tests.helpers import random_positions

    pos = random_positions(n, seed=seed)
    pl = np.zeros((n, 2, 42), dtype=np.float32)
    for i, (m, v) in enumerate(pos):
        for b in range(42):
            pl[i, 0, b] = (v >> b) & 1
            pl[i, 1, b] = ((m & ~v) >> b) & 1
    return torch.from_numpy(pl.reshape(n, 2, 6, 7))


def _ref_tower(model, x):
    """fp32 reference with the SAME bf16-rounded weights (so only accumulation/rounding of
    activations differs)."""
    import torch.nn.functional as F
    from c4a0_amd.nn import _fold_bn

    r = lambda t: t.detach().float().bfloat16().float()
    conv0 = model.conv[0]
    y = F.conv2d(x, r(conv0.weight), conv0.bias.detach().float(), padding=1)
    for blk in list(model.conv)[1:]:
        c1, c2, bn = blk.block[0], blk.block[1], blk.block[2]
        w2, b2 = _fold_bn(c2.weight, c2.bias, bn)
        t = F.conv2d(y, r(c1.weight), c1.bias.detach().float(), padding=1)
        t = F.conv2d(t, r(w2), b2, padding=1)
        y = y + F.relu(t)
    return y


@pytest.mark.parametrize("channels,blocks,n", [(32, 1, 37), (32, 4, 1003), (32, 4, 4096), (64, 2, 100), (64, 8, 512)])
def test_hip_conv_tower_vs_fp32_reference(channels, blocks, n):
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(channels + blocks)
    model = ConnectFourNet(ModelConfig(blocks, channels, 2, 2)).eval()
    g = torch.Generator().manual_seed(5)
    for m in model.modules():  # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    x = _positions(n)
    net = InferenceNet(model, dev, dtype=torch.bfloat16, hip_tower=True)
    got = net.tower(x.to(dev).bfloat16()).float().cpu().reshape(n, 42, channels).permute(0, 2, 1).reshape(n, channels, 6, 7)
    with torch.no_grad():
        want = _ref_tower(model, x)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    # bf16 activations between layers: 8 bits of mantissa per layer, 1 + 2*blocks layers
    assert err <= 0.02 * scale * (1 + blocks) ** 0.5, (err, scale)
    # and it is not trivially zero / permuted: correlation with the reference
    assert torch.corrcoef(torch.stack([got.flatten(), want.flatten()]))[0, 1] > 0.999


def test_inference_net_hip_vs_torch_paths_and_reference_fixture():
    """Whole evaluator on the device: HIP-tower path vs PyTorch-conv path vs fp32 module."""
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    model = ConnectFourNet(ModelConfig(4, 32, 4, 2)).eval()
    x = _positions(777, seed=9)
    with torch.no_grad():
        lp_ref, qp_ref, qn_ref = model(x)
    for tower in (True, False):
        net = InferenceNet(model, dev, dtype=torch.bfloat16, hip_tower=tower)
        lp, q = net(x.to(dev).bfloat16())
        lp, q = lp.cpu(), q.cpu()
        assert lp.dtype == torch.float32 and q.shape == (777, 2)
        assert (lp - lp_ref).abs().max() < 0.05, (tower, (lp - lp_ref).abs().max())     # bf16 weights + activations
        assert (q[:, 0] - qp_ref).abs().max() < 0.05 and (q[:, 1] - qn_ref).abs().max() < 0.05
        assert torch.allclose(lp.exp().sum(1), torch.ones(777), atol=1e-4)


def test_graphed_evaluator_matches_eager_and_drives_a_session():
    from c4a0_amd.nn import ConnectFourNet, GraphedEvaluator, InferenceNet, ModelConfig
    from c4a0_amd.session import DeviceSession

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    s = DeviceSession(64, 8, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games([(i, 0, 0) for i in range(100)])
    s.bind()
    s.start()
    ge = GraphedEvaluator(net, s.planes, s.logprobs, s.q)
    for _ in range(5):
        s.evaluate(ge)
        lp_e, q_e = net(s.planes)
        assert torch.equal(lp_e, s.logprobs) and torch.equal(q_e, s.q)
        s.step()
    steps = 5
    while s.counters()["games_done"] < 100 and steps < 20000:
        for _ in range(64):
            s.evaluate(ge)
            s.step()
        steps += 64
    c = s.counters()
    assert c["games_done"] == 100 and c["error"] == 0
    recs = s.drain_samples()
    assert len(recs) == c["samples"] and set(np.unique(recs["game_id"]).tolist()) == set(range(100))
    s.close()


@pytest.mark.parametrize("features,n", [(1344, 1), (1344, 777), (1344, 4096), (2688, 530), (320, 100)])
def test_head_output_kernel_vs_fp32_reference(features, n):
    """c4_head_out_bf16 (policy Linear+LogSoftmax, value Linear+Tanh; nn.py:84-85, 98-99): the MFMA
    form (features = 42 C) and the dot-product form (other sizes) against fp32 PyTorch on the same
    bf16 operands, with strided hidden activations as the merged first layer produces them."""
    import ctypes as C
    from c4a0_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(features + n)
    both = (torch.randn(n, 2 * features, generator=g) * 0.5).bfloat16().to(dev)   # [policy hidden | value hidden]
    hp, hv = both[:, :features], both[:, features:]
    wp = (torch.randn(7, features, generator=g) / features ** 0.5).bfloat16().to(dev)
    wv = (torch.randn(2, features, generator=g) / features ** 0.5).bfloat16().to(dev)
    bp = torch.randn(7, generator=g).to(dev)
    bv = torch.randn(2, generator=g).to(dev)
    lp = torch.full((n, 7), float("nan"), device=dev)
    q = torch.full((n, 2), float("nan"), device=dev)
    _lib.check(L.c4_head_out_bf16(C.c_void_p(hp.data_ptr()), C.c_void_p(hv.data_ptr()), C.c_void_p(wp.data_ptr()),
                                  C.c_void_p(wv.data_ptr()), C.c_void_p(bp.data_ptr()), C.c_void_p(bv.data_ptr()),
                                  n, features, hp.stride(0), hv.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(q.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want_lp = torch.log_softmax(hp.float() @ wp.float().T + bp, dim=1)
    want_q = torch.tanh(hv.float() @ wv.float().T + bv)
    # f32 accumulation of exact bf16 products on both sides: only the summation order differs
    assert (lp - want_lp).abs().max().item() < 2e-4, (lp - want_lp).abs().max().item()
    assert (q - want_q).abs().max().item() < 2e-4, (q - want_q).abs().max().item()


def test_tower_workgroup_shapes_compute_the_same_bits():
    """c4_conv_tower_bf16's config picks what a workgroup owns (16 or 8 boards, 8 or 12 wavefronts), never a
    board's arithmetic: the same features bit for bit, at sizes on both sides of the automatic choice's cut,
    with a ragged last workgroup; an unknown config is refused."""
    import ctypes as C
    from c4a0_amd._lib import lib
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = ConnectFourNet(ModelConfig(2, 32, 2, 2)).eval()
    net = InferenceNet(model, dev, dtype=torch.bfloat16, hip_tower=True)
    for n in (1, 37, 1280, 1283, 2048):
        x = (torch.rand(n, 2, 6, 7) < 0.3).to(dev).bfloat16()
        net.tower_config = 0
        want = net.tower(x)
        for cfg in (1, 2, 3):
            net.tower_config = cfg
            assert torch.equal(net.tower(x).view(torch.int16), want.view(torch.int16)), (n, cfg)
    net.tower_config, net.latency_mode = 0, True       # alone on the device: the 8-board shape by itself
    assert torch.equal(net.tower(x).view(torch.int16), want.view(torch.int16))
    out = torch.empty((1, 42 * 32), dtype=torch.bfloat16, device=dev)
    rc = lib().c4_conv_tower_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(net.tw0.data_ptr()), C.c_void_p(net.tw.data_ptr()), C.c_void_p(net.tbias.data_ptr()),
                                  1, 32, 2, C.c_void_p(out.data_ptr()), 9, None)
    assert rc != 0 and b"config" in lib().c4_last_error_string()
