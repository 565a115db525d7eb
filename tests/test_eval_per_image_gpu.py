"""Inference on the three-product bodies with one magnitude slot per IMAGE (rd_quant_next_img, r06; -m gpu).

The tiled sweep of lib/evaluation.py:460-567 promises the same raster however the tiles are batched or sharded, so a tile's result
must not depend on the tiles that share its batch -- which one scale per operand TENSOR (the training form of split2h) would
break.  Per-image slot arrays keep the promise at three products per multiply: a block scales its activation operand by its own
image's maximum.  lib/UNet.py:196-246 in eval mode is the forward being computed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed=3, depth=5):
    import resdepth_amd
    from resdepth_amd import _lib
    if _lib.products() != 3:
        pytest.skip("per-image slots belong to the split2h mode")
    torch.manual_seed(seed)
    m = resdepth_amd.UNet(n_input_channels=3, start_kernel=64, depth=depth).cuda()
    with torch.no_grad():                                   # running statistics that are not the identity
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
    return m.eval()


def _fwd(m, x, **flags):
    keep = {k: getattr(m, k) for k in flags}
    for k, v in flags.items():
        setattr(m, k, v)
    try:
        with torch.no_grad():
            return m(x).clone()
    finally:
        for k, v in keep.items():
            setattr(m, k, v)


def test_a_tile_does_not_depend_on_its_batch_mates_and_takes_the_three_product_bodies():
    m = _model()
    g = torch.Generator(device="cuda").manual_seed(11)
    t0 = torch.randn(1, 3, 256, 256, device="cuda", generator=g)
    mates_a = torch.randn(7, 3, 256, 256, device="cuda", generator=g)
    mates_b = torch.randn(7, 3, 256, 256, device="cuda", generator=g) * 300.0      # very different magnitudes in the same batch
    out_a = _fwd(m, torch.cat([t0, mates_a]))
    out_b = _fwd(m, torch.cat([t0, mates_b]))
    assert torch.equal(out_a[0], out_b[0])                  # per-image scales: bit for bit the same tile
    # ... wherever it sits in the batch
    out_c = _fwd(m, torch.cat([mates_b[:3], t0, mates_a[:4]]))
    assert torch.equal(out_a[0], out_c[3])
    # ... and whatever the batch size (the ragged last batch of a sweep): three or six products is decided by the layer's shape
    more = torch.cat([mates_b, mates_a * 0.01, mates_a[:2] * 7.0])
    for nb in (1, 3, 17):
        out_n = _fwd(m, torch.cat([t0, more[:nb - 1]]) if nb > 1 else t0)
        assert torch.equal(out_a[0], out_n[0]), nb
    # the six-product form of r01-r05 computes the same forward: close, and NOT the same bits (the three-product bodies did run)
    six = _fwd(m, torch.cat([t0, mates_a]), eval_per_image=False)
    assert not torch.equal(six, out_a)
    scale = float(six.abs().max())
    assert float((six - out_a).abs().max()) <= 2e-5 * scale
    # one scale per tensor (fast_eval) is what breaks the promise: the big mates change tile 0's bits
    fa = _fwd(m, torch.cat([t0, mates_a]), fast_eval=True)
    fb = _fwd(m, torch.cat([t0, mates_b]), fast_eval=True)
    assert float((fa[0] - six[0]).abs().max()) <= 2e-5 * scale
    assert not torch.equal(fa[0], fb[0])


def test_per_image_forward_against_the_fp64_oracle_of_the_same_network():
    """Eval forward of a depth-4 net on 128 x 128 tiles against the oracle (oracle/unet_oracle.py, lib/UNet.py:196-246) in fp64."""
    from oracle import unet_oracle as O
    m = _model(seed=5, depth=4)
    x = torch.randn(4, 3, 128, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    out = _fwd(m, x)
    six = _fwd(m, x, eval_per_image=False)
    sd64 = {k: (v.double().cpu() if v.is_floating_point() else v.cpu()) for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.forward(sd64, x.double().cpu(), O.Spec(n_input_channels=3, start_kernel=64, depth=4), training=False)
    scale = float(ref.abs().max())
    e3, e6 = float((out.double().cpu() - ref).abs().max()) / scale, float((six.double().cpu() - ref).abs().max()) / scale
    assert e3 <= 2e-5 and e6 <= 2e-5, (e3, e6)
    assert e3 <= 4 * e6 + 1e-6, (e3, e6)                    # the three-product form is not the less accurate one by any margin that matters


def test_slots_nobody_wrote_send_the_consumer_to_the_six_product_body():
    """A per-image slot array that no producer filled reads as zero = magnitude unknown (quant_select): same bits as no slot."""
    from resdepth_amd import _lib, ops
    if _lib.products() != 3:
        pytest.skip("split2h only")
    torch.manual_seed(0)
    x = torch.randn(4, 32, 32, 64, device="cuda")
    w = torch.randn(128, 64, 3, 3, device="cuda") * 0.05
    shift = torch.zeros(128, device="cuda")
    wf = ops.pack_conv3x3_weight_folded(w, torch.ones(128, device="cuda"))
    plain, _ = ops.conv3x3_fwd_act(x, wf, shift, 0.0)
    with _lib.AmaxPool(x.device, slots=4, per_image=4) as pool:
        xs = _lib.tag(x.clone(), pool.take())               # tagged, never written
        got, _ = ops.conv3x3_fwd_act(xs, wf, shift, 0.0)
        assert torch.equal(got, plain)
        # the launch committed ITS output's maxima per image
        sl = _lib.slot_of(got).view(4, -1)
        want = got.abs().amax(dim=(1, 2, 3))
        assert torch.equal(sl.max(dim=1).values.contiguous().view(torch.float32), want)      # IEEE bit patterns of |x|, integer max
