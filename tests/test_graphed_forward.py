"""``CRAFT.capture`` / ``GraphedForward``: the inference pass recorded as one hipGraph (craft_amd/network.py) replays the SAME kernels in the
same order -- held here to the eager pass (two separate passes agree up to the summation order of the double-precision statistics
atomics: 1e-4 px) and to the reference's captures."""
import pytest
import torch

from craft_amd import CRAFT, GraphedForward, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from golden_util import Golden
from test_hip_e2e import build


def test_capture_refuses_cpu_inputs_and_training_mode():
    model = CRAFT(default_args()).eval()
    im = torch.zeros(1, 3, 128, 160)
    with pytest.raises(RuntimeError, match="GPU"):
        model.capture(im, im)
    model.train()
    with pytest.raises(RuntimeError, match="eval"):
        GraphedForward(model, im, im)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_replay_matches_eager_and_the_reference_capture(device, precision):
    g = Golden("canon_128x256_T4")
    model = build(g, device, precision)
    im1, im2 = (t.to(device) for t in g.images())
    with torch.no_grad():
        lo_e, ups_e = model(im1, im2, iters=4, test_mode=2)
    lo_e, ups_e = lo_e.clone(), [u.clone() for u in ups_e]
    gf = model.capture(im1, im2, iters=4, test_mode=2)
    for rep in range(3):
        lo, ups = gf(im1, im2)
        assert (lo - lo_e).abs().max().item() < 1e-4
        assert len(ups) == 4 and all((a - b).abs().max().item() < 1e-4 for a, b in zip(ups, ups_e)), f"replay {rep}"
    if precision == "fp32":
        g.check("flow_lo", lo, 1e-3, 1e-3)
        for it, p in enumerate(ups):
            g.check(f"up{it}", p, 1e-3, 3e-3)
    assert gf.replays == 3


@pytest.mark.gpu
def test_replay_follows_new_inputs_and_flow_init(device):
    """The graph reads its static input buffers: other pairs (and another warm start) of the captured shape give what the eager pass gives."""
    g = Golden("canon_b2_128x192_T3_init")
    model = build(g, device, "mixed")
    im1, im2 = (t.to(device) for t in g.images())
    fi = g.flow_init().to(device)
    gf = model.capture(im1, im2, iters=3, flow_init=fi, test_mode=1)
    a1, a2, _ = synth_pair(2, 128, 192, seed=7)
    a1, a2 = a1.to(device), a2.to(device)
    fi2 = (fi * -0.5 + 0.25).contiguous()
    for x1, x2, f in ((a1, a2, fi2), (im1, im2, fi), (a2, a1, fi2)):
        lo, up = gf(x1, x2, f)
        with torch.no_grad():
            lo_e, up_e = model(x1, x2, iters=3, flow_init=f, test_mode=1)
        assert (lo - lo_e).abs().max().item() < 1e-4 and (up - up_e).abs().max().item() < 1e-4
    with pytest.raises(ValueError, match="flow_init"):
        gf(im1, im2)
    with pytest.raises(ValueError, match="shape"):
        gf(im1[:1], im2[:1], fi[:1])


@pytest.mark.gpu
def test_changed_weights_are_recaptured(device):
    """The graph holds the addresses of the weight operands packed at capture time; a parameter changed in place (an optimizer step, a
    checkpoint load) must not be served from the old packs."""
    model = CRAFT(default_args(hip_precision="mixed"))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=5), strict=True)
    model = model.to(device).eval()
    im1, im2, _ = synth_pair(1, 128, 160, seed=3)
    im1, im2 = im1.to(device), im2.to(device)
    gf = model.capture(im1, im2, iters=2)
    lo0, up0 = (t.clone() for t in gf(im1, im2))
    with torch.no_grad():
        model.update_block.gru.convz1.weight.mul_(1.5)
        model.update_block.flow_head.conv2.weight.mul_(0.5)
    first = gf.graph
    lo1, up1 = gf(im1, im2)
    assert gf.graph is not first, "weights changed: the pass must be recorded again"
    with torch.no_grad():
        lo_e, up_e = model(im1, im2, iters=2, test_mode=1)
    assert (up1 - up_e).abs().max().item() < 1e-4 and (lo1 - lo_e).abs().max().item() < 1e-4
    assert (up1 - up0).abs().max().item() > 1e-3, "the edit must have changed the prediction"
    again = gf.graph
    gf(im1, im2)
    assert gf.graph is again
    # the precision policy is part of what was recorded
    model.args.hip_precision = "fp32"
    lo2, up2 = gf(im1, im2)
    assert gf.graph is not again
    with torch.no_grad():
        lo_f, up_f = model(im1, im2, iters=2, test_mode=1)
    assert (up2 - up_f).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_capture_with_batch_sliced_streams(device):
    """args.hip_streams = 2 (the auto rule for 6..12 pairs) slices the eager loop over two streams; under a capture the batch stays in one
    piece (network.py: hipStreamEndCapture of the two-slice pass crashed inside the runtime at 448x1024 x 4) and the replay equals the
    sliced eager pass."""
    g = Golden("canon_b2_128x192_T3_init")
    model = build(g, device, "mixed")
    model.args.hip_streams = 2
    im1, im2 = (t.to(device) for t in g.images())
    with torch.no_grad():
        lo_e, up_e = (t.clone() for t in model(im1, im2, iters=3, test_mode=1))
    gf = model.capture(im1, im2, iters=3, test_mode=1)
    for _ in range(2):
        lo, up = gf(im1, im2)
        assert (lo - lo_e).abs().max().item() < 1e-4 and (up - up_e).abs().max().item() < 1e-4
