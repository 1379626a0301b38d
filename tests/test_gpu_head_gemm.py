"""c4_linear_bf16 (hidden layers of the heads, reference src/c4a0/nn.py:75-100) on the GPU:
numerics against an fp32 PyTorch reference of the same op, and the property the kernel exists for --
a row's result is a function of that row and the weights only (bit-identical whatever the batch
size, the row's index or the tile configuration)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = list(range(1, 60))   # 29-34: round-4 tiles, 35-45: wave-specialised forms (loading wavefronts), 46-47: staggered wavefront halves


def _linear(L, x, w, b32, relu=1, config=0, out=None):
    from c4a0_amd._lib import check
    m, n, k = x.shape[0], w.shape[0], w.shape[1]
    y = out if out is not None else torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    check(L.c4_linear_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b32.data_ptr()), C.c_void_p(y.data_ptr()),
                           m, n, k, x.stride(0), y.stride(0), relu, config, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


@pytest.fixture(scope="module")
def ops():
    from c4a0_amd import _lib
    torch.manual_seed(7)
    dev = torch.device("cuda:0")
    k, n = 1344, 2688
    x = torch.randn(2048, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    return _lib.lib(), x, w, b


@pytest.mark.parametrize("config", [0] + CONFIGS)
@pytest.mark.parametrize("m", [1, 77, 300, 2048])
def test_matches_fp32_reference(ops, config, m):
    L, x, w, b = ops
    for relu in (1, 0):
        y = _linear(L, x[:m], w, b, relu=relu, config=config)
        ref = x[:m].float() @ w.float().t() + b
        if relu:
            ref = torch.relu(ref)
        # f32 accumulation of bf16 products, one rounding to bf16 at the end: within one bf16 ulp of the fp32 result
        torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -7)


def test_a_row_is_a_function_of_the_row_only(ops):
    """Batch size, row index and tile configuration do not change a single bit."""
    L, x, w, b = ops
    full = _linear(L, x, w, b, config=1)
    for config in CONFIGS:
        assert torch.equal(_linear(L, x, w, b, config=config), full), f"config {config} differs from config 1"
        for m in (1, 5, 77, 300, 1000):
            assert torch.equal(_linear(L, x[:m], w, b, config=config), full[:m]), f"config {config}, batch of {m}"
    perm = torch.randperm(x.shape[0], device=x.device)
    assert torch.equal(_linear(L, x[perm].contiguous(), w, b), full[perm])
    # a batch made of one position repeated: every row equal
    rep = x[123:124].expand(333, -1).contiguous()
    y = _linear(L, rep, w, b)
    assert torch.equal(y, full[123:124].expand(333, -1))


def test_strided_operands_and_narrow_layers(ops):
    """x as a column range of a wider tensor (the merged first layer's halves), N = K = 1344."""
    L, x, w, b = ops
    h = _linear(L, x[:512], w, b)                       # [512, 2688]
    w2 = w[:1344, :1344].contiguous()
    for lo in (0, 1344):
        part = h[:, lo:lo + 1344]
        assert part.stride(0) == 2688
        y = _linear(L, part, w2, b[:1344].contiguous())
        ref = torch.relu(part.float() @ w2.float().t() + b[:1344])
        torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -7)
        assert torch.equal(y, _linear(L, part.contiguous(), w2, b[:1344].contiguous()))


@pytest.mark.parametrize("n", [192, 384, 768, 1536])
def test_tile_grids_one_xcd_rectangle_wide(ops, n):
    """N in {192, 384, 768, 1536}: an XCD's rectangle of tiles is ONE tile wide (rn == 1), where round 4's division by
    multiplication sent blocks past N.  Several row tiles, every configuration whose tile divides N."""
    L, x, w, b = ops
    wn, bn = w[:n].contiguous(), b[:n].contiguous()
    for m in (2048, 700):
        guard = torch.full((m + 8, n), 7.0, dtype=torch.bfloat16, device=x.device)   # rows past m must stay untouched
        ref = torch.relu(x[:m].float() @ wn.float().t() + bn)
        for config in [0] + CONFIGS:
            guard.fill_(7.0)
            y = _linear(L, x[:m], wn, bn, config=config, out=guard[:m])   # every tile width (192, 96, 64) divides a multiple of 192
            torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -7)
            assert bool((guard[m:] == 7.0).all()), f"config {config} wrote past the last row"


def test_bad_arguments_are_refused(ops):
    from c4a0_amd._lib import C4Error
    L, x, w, b = ops
    with pytest.raises(C4Error):
        _linear(L, x[:8, :1000].contiguous(), w[:, :1000].contiguous(), b)     # K not a multiple of 64
    with pytest.raises(C4Error):
        _linear(L, x[:8], w[:100].contiguous(), b[:100].contiguous())          # N not a multiple of 192
    with pytest.raises(C4Error):
        _linear(L, x[:8], w, b, config=77)


@pytest.mark.parametrize("blocks,channels", [(4, 32), (2, 64)])
def test_evaluator_is_batch_invariant(blocks, channels):
    """The whole bf16 evaluator (tower + hand GEMMs + output kernel): a position's outputs do not depend
    on the batch it is evaluated in (what makes play_games independent of mode and placement).  Both channel
    counts: the automatic tile choice differs with K (256 x 192 above 1 024 rows at K = 2 688) and with the batch."""
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(blocks, channels, 4, 2)), dev, dtype=torch.bfloat16)
    assert net.gemm == "hip"
    planes = (torch.rand(4096, 2, 6, 7, device=dev) < 0.3).to(torch.bfloat16)
    lp, q = net(planes)
    lp, q = lp.clone(), q.clone()
    for m in (1, 3, 100, 1000, 1281, 2048):
        lp_m, q_m = net(planes[:m].contiguous())
        assert torch.equal(lp_m, lp[:m]) and torch.equal(q_m, q[:m]), f"batch of {m}"
    perm = torch.randperm(4096, device=dev)[:1500]
    lp_p, q_p = net(planes[perm].contiguous())
    assert torch.equal(lp_p, lp[perm]) and torch.equal(q_p, q[perm])
