"""Host-side pieces of the training step -- CPU: the OneCycle schedule against torch's own
scheduler, the flat parameter / gradient views, the AdamW state layout, and the one-collective gradient exchange under
gloo with two ranks."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from craft_amd.train import FlatAdamW, OneCycleLR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("num_steps,lr", [(1000, 4e-4), (120, 1.25e-4)])
def test_onecycle_matches_torch(num_steps, lr):
    # exactly what fetch_optimizer builds (train.py:76-85)
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([p], lr=lr, weight_decay=1e-5, eps=1e-8)
    ref = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=lr, total_steps=num_steps + 100, pct_start=0.05,
                                              cycle_momentum=False, anneal_strategy="linear")
    ours = OneCycleLR(lr, num_steps + 100, pct_start=0.05)
    for _ in range(num_steps + 99):
        assert ours.get_last_lr()[0] == pytest.approx(ref.get_last_lr()[0], rel=1e-12, abs=1e-18)
        opt.step()
        ref.step()
        ours.step()
    sd = ours.state_dict()
    again = OneCycleLR(lr, num_steps + 100)
    again.load_state_dict(sd)
    assert again.get_last_lr() == ours.get_last_lr()


def test_flat_views_and_state_layout():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Linear(5, 2))
    before = [p.detach().clone() for p in net.parameters()]
    opt = FlatAdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, eps=1e-8)
    assert opt.n_params == sum(p.numel() for p in net.parameters()) and opt.n_params <= opt.numel < opt.n_params + 32 * 4
    for p, b, off in zip(net.parameters(), before, opt.offsets):
        assert off % 32 == 0                                                     # 128-byte aligned starts (16-byte vector loads)
        assert torch.equal(p.detach(), b)                                       # values survive the re-pointing
        assert p.data.data_ptr() == opt.flat[off:].data_ptr()                   # parameters are views of the flat buffer
        assert p.grad.data_ptr() == opt.flat_grad[off:].data_ptr()              # and so are their gradients
    # autograd accumulates straight into the flat gradient buffer
    net(torch.randn(1, 3, 7, 7)).sum().backward()
    assert opt.flat_grad.abs().sum() > 0
    # the optimizer state has torch.optim.AdamW's layout (what the reference's checkpoints store, train.py:139)
    ref = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, eps=1e-8)
    ref.step()
    sd, rsd = opt.state_dict(), ref.state_dict()
    assert set(sd.keys()) == set(rsd.keys()) and sd["param_groups"][0]["params"] == rsd["param_groups"][0]["params"]
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["exp_avg"].shape == before[0].shape
    opt.load_state_dict(rsd)                                                     # a torch AdamW state loads into ours
    assert opt.step_count == 1 and torch.equal(opt.state_dict()["state"][1]["exp_avg"], rsd["state"][1]["exp_avg"])


def test_load_grads_copies_captured_gradients_and_clears_stale_views():
    """FlatAdamW.load_grads (what Trainer.step does with torch.autograd.grad's result): equal to zero_grad() + backward(), non-contiguous
    gradients included; a parameter that loses its gradient gets its view zeroed; p.grad keeps pointing into the flat buffer."""
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Flatten(), torch.nn.Linear(100, 2), torch.nn.Linear(2, 3))
    opt = FlatAdamW(net.parameters(), lr=1e-3)
    x = torch.randn(2, 3, 7, 7)
    opt.zero_grad()
    net(x).sum().backward()
    want = opt.flat_grad.clone()
    opt.flat_grad.fill_(7.0)                       # (load_grads does not rely on a zeroed buffer for the views it writes)
    pad = torch.ones_like(opt.flat_grad, dtype=torch.bool)
    for p, off in zip(opt.params, opt.offsets):
        pad[off:off + p.numel()] = False
    opt.flat_grad[pad] = 0.0                       # (the alignment padding is zero from the allocation and stays so)
    grads = list(torch.autograd.grad(net(x).sum(), opt.params, allow_unused=True))
    grads[0] = grads[0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)          # a strided view, as the kernels' accumulators return
    opt.load_grads(grads)
    assert torch.equal(opt.flat_grad, want)
    for p, off in zip(opt.params, opt.offsets):
        assert p.grad.data_ptr() == opt.flat_grad[off:].data_ptr()
    # the last layer drops out of the graph: its views are cleared, the others rewritten
    g2 = list(torch.autograd.grad(net[:3](x).sum(), opt.params, allow_unused=True))
    assert g2[-1] is None and g2[-2] is None
    opt.load_grads(g2)
    n_last = opt.params[-1].numel() + opt.params[-2].numel()
    assert float(opt.grad_views()[-1].abs().sum()) == 0.0 and float(opt.grad_views()[-2].abs().sum()) == 0.0 and n_last > 0
    assert float(opt.grad_views()[0].abs().sum()) > 0
    with pytest.raises(ValueError):
        opt.load_grads([torch.zeros(1)] + g2[1:])


def test_tiled_probability_layout_helpers():
    """ops.probs_tiled / probs_rowmajor / vt_stride (the CRAFT_P_TILED layout of include/craft_hip.h) on the CPU: element (i, j) at
    (i >> 5) * 32 * ldp + (j >> 6) * 2048 + (i & 31) * 64 + (j & 63), key extent rounded to 64, V^T's to 32."""
    from craft_amd import ops
    B, M, N = 2, 3, 100
    P = torch.arange(B * M * N * N, dtype=torch.float32).view(B, M, N, N)
    Pp = torch.zeros(B, M, N, ops.round_up(N, 32))
    Pp[..., :N] = P
    T = ops.probs_tiled(Pp, fill=-1.0)
    assert T.craft_tiled and T.craft_n == N and tuple(T.shape) == (B, M, 128, 128) and ops.vt_stride(T) == 128 and ops.vt_stride(Pp) == 128
    flat = T[1, 2].reshape(-1)
    ldp = 128
    for i, j in ((0, 0), (31, 63), (32, 64), (99, 99), (45, 70), (64, 5)):
        assert float(flat[(i >> 5) * 32 * ldp + (j >> 6) * 2048 + (i & 31) * 64 + (j & 63)]) == float(P[1, 2, i, j])
    assert float(flat[(3 * 32) * ldp + 0 * 2048 + 4 * 64 + 0]) == -1.0            # row 100: padding row keeps the fill
    assert float(flat[(0 >> 5) * 32 * ldp + (100 >> 6) * 2048 + 0 * 64 + (100 & 63)]) == 0.0     # column 100 of a real row: zero
    assert torch.equal(ops.probs_rowmajor(T), P)
    sl = ops.probs_slice(T, 1, 2)
    assert sl.craft_tiled and sl.craft_n == N and torch.equal(ops.probs_rowmajor(sl), P[1:2])
    N2 = 40                                        # round_up(N, 32) = 64 = round_up(N, 64); N = 70: 96 vs 128
    assert ops.vt_stride(ops.probs_tiled(torch.zeros(1, 1, 70, 96))) == 96


WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from craft_amd.train import FlatAdamW
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)
    net = torch.nn.Linear(6, 3)
    opt = FlatAdamW(net.parameters())
    opt.flat_grad.copy_(torch.arange(opt.numel, dtype=torch.float32) * (rank + 1))
    mul = opt.allreduce_grads()
    # ONE collective over the whole flat buffer: sum over ranks, 1/world returned for the update
    assert mul == 1.0 / world
    assert torch.equal(opt.flat_grad, torch.arange(opt.numel, dtype=torch.float32) * 3), opt.flat_grad
    assert torch.equal(net.weight.grad.reshape(-1), opt.flat_grad[:18])
    dist.barrier()
    dist.destroy_process_group()
    print("ok", rank)
""")


def test_gradient_allreduce_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:                      # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_reference_style_full_checkpoint_loads(tmp_path):
    """A checkpoint as the reference's trainer writes it (train.py:132-145): numpy scalars in 'logger', OneCycleLR state in
    'lr_scheduler'.  The safe unpickler with numpy globals allow-listed must read it; an arbitrary pickled
    object is refused unless trusted=True."""
    import pickle
    import numpy as np
    from craft_amd import CRAFT, default_args
    from craft_amd.utils import load_checkpoint, read_checkpoint
    model = CRAFT(default_args())
    sd = {"module." + k: v for k, v in model.state_dict().items()}
    ck = {"model": sd, "optimizer": {"state": {}, "param_groups": [{"lr": 1e-4, "params": [0]}]},
          "lr_scheduler": {"total_steps": 100, "last_epoch": 3, "_last_lr": [1e-4], "anneal_func": "linear"},
          "logger": {"train_epe_list": [np.float64(1.5), np.mean(np.array([1.0, 2.0], dtype=np.float32))], "total_steps": 3}}
    path = str(tmp_path / "ref_style.pth")
    torch.save(ck, path)
    res = load_checkpoint(CRAFT(default_args()), path, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert read_checkpoint(path)["logger"]["total_steps"] == 3

    import argparse
    ck["logger"]["odd"] = argparse.Namespace(a=1)            # something the safe unpickler must refuse
    torch.save(ck, path)
    with pytest.raises(pickle.UnpicklingError):
        read_checkpoint(path)
    assert "odd" in read_checkpoint(path, trusted=True)["logger"]


def test_zero_pool_serves_a_recorded_sequence_and_falls_back():
    """hip.ZeroPool (the per-step source of the small zero-initialised buffers): the first step records, later steps hand out disjoint
    zeroed pieces of ONE allocation in the recorded order; a deviating request falls back to plain allocations and the plan is re-learned."""
    from craft_amd import hip
    pool = hip.ZeroPool()
    dev = torch.device("cpu")
    seq = [((3, 5), torch.float32), ((7,), torch.float64), ((2, 2, 2), torch.float32)]

    def step(requests):
        pool.begin(dev)
        return [pool.zeros(s, dev, dt) for s, dt in requests]

    first = step(seq)                                       # recording: plain tensors
    assert pool.flat is None and all(float(t.abs().sum()) == 0 for t in first)
    second = step(seq)                                      # served from the flat buffer
    assert pool.flat is not None and pool.ok
    for t, (s, dt) in zip(second, seq):
        assert tuple(t.shape) == s and t.dtype == dt and float(t.abs().sum()) == 0
        assert t.untyped_storage().data_ptr() == pool.flat.untyped_storage().data_ptr()
    second[0].fill_(1.0)                                    # pieces are disjoint ...
    assert float(second[1].abs().sum()) == 0 and float(second[2].abs().sum()) == 0
    third = step(seq)                                       # ... and a new step starts from zeros in a NEW allocation
    assert float(third[0].abs().sum()) == 0 and third[0].untyped_storage().data_ptr() != second[0].untyped_storage().data_ptr()
    other = [((3, 5), torch.float32), ((9,), torch.float32), ((2, 2, 2), torch.float32)]
    fourth = step(other)                                    # deviation at the 2nd request: fallback from there on
    assert not pool.ok and fourth[1].shape == (9,) and float(fourth[2].abs().sum()) == 0
    fifth = step(other)                                     # the new sequence has been learned
    assert pool.ok and fifth[1].untyped_storage().data_ptr() == pool.flat.untyped_storage().data_ptr()
    # outside a training step hip.zeros is torch.zeros
    hip.set_zero_pool(None)
    assert float(hip.zeros((4,), dev).sum()) == 0


def test_carray_types_are_cached():
    import ctypes
    from craft_amd import hip
    a, b = hip.carray(ctypes.c_long, [1, 2, 3]), hip.carray(ctypes.c_long, [4, 5, 6])
    assert type(a) is type(b) and list(b) == [4, 5, 6] and type(hip.carray(ctypes.c_long, [1])) is not type(a)


def test_training_driver_accepts_the_reference_command_lines():
    """craft_amd.train_main.parse takes the argument lists of the reference's shipped training scripts (train-craft-f2full.sh, train-gma.sh,
    train-craft-f2full-gma.sh: everything after `python3 train.py`) for the stages whose datasets have a walker here."""
    from craft_amd import default_args
    from craft_amd.train_main import parse
    lines = [
        "--name craft-chairs --stage chairs --validation chairs --output results/chairs/craft-f2full --num_steps 120000 --lr 0.00025 --image_size 368 496 "
        "--wdecay 0.0001 --gpus 0 1 --batch_size 8 --val_freq 10000 --print_freq 100 --mixed_precision --craft --f2 full --setrans",
        "--name craft-sintel --stage sintel --validation sintel --output results/sintel/craft-f2full --restore_ckpt results/things/craft-f2full/craft-things.pth "
        "--num_steps 120000 --lr 0.000125 --image_size 368 768 --wdecay 0.00001 --gamma 0.85 --batch_size 6 --val_freq 10000 --print_freq 100 --mixed_precision "
        "--craft --f2 full --setrans",
        "--name gma-kitti --stage kitti --validation kitti --output results/kitti/gma --restore_ckpt results/sintel/gma/gma-sintel.pth --num_steps 50000 "
        "--lr 0.000125 --image_size 288 960 --wdecay 0.00001 --gamma 0.85 --gpus 0 1 --batch_size 6 --val_freq 10000 --print_freq 100 --mixed_precision",
    ]
    ns = [parse(l.split()) for l in lines]
    assert ns[0].stage == "chairs" and ns[0].image_size == [368, 496] and ns[0].craft and ns[0].use_setrans and ns[0].f2trans == "full" and ns[0].mixed_precision
    assert ns[1].gamma == 0.85 and ns[1].restore_ckpt.endswith("craft-things.pth") and ns[1].lr == 0.000125
    assert not ns[2].craft and not ns[2].use_setrans and ns[2].gpus == [0, 1]
    # every model switch the parser produces is a field CRAFT(args) reads, with the reference's defaults (train.py:313-404)
    d = parse(["--stage", "chairs"])
    known = vars(default_args())
    for k in ("f2trans", "f1trans", "inter_num_modes", "intra_num_modes", "f2_num_modes", "inter_qk_have_bias", "inter_pos_code_weight", "intra_pos_code_weight",
              "f2_pos_code_weight", "f2_attn_mask_radius", "pos_bias_radius", "num_heads", "corr_radius", "dropout"):
        assert k in known and k in vars(d)
    assert (d.f2trans, d.lr, d.wdecay, d.batch_size, d.image_size, d.iters, d.clip, d.gamma) == ("full", 0.00002, 0.00005, 6, [384, 512], 12, 1.0, 0.8)


def test_backward_operand_roles_of_the_policies():
    """hip.Precision roles wgx / wgy / dxw / sbw: `mixed` and `train_bf16attn` run the backward's non-propagating products
    on single fp16 planes -- under an announced loss scale only; without one (a bare loss.backward()) every fp16 role is promoted: the
    forward roles to f16x3, the backward roles to "the layer's mode"."""
    from craft_amd.hip import PREC_BF16, PREC_F16, PREC_F16X3, Precision
    from craft_amd.train_forward import training_precision
    m = Precision.parse("mixed")
    assert (m.proj, m.score, m.pv, m.conv, m.enc) == (PREC_F16X3, PREC_F16X3, PREC_F16, PREC_F16X3, PREC_F16X3)
    assert (m.wgx, m.wgy, m.dxw, m.sbw) == (PREC_F16,) * 4
    b = Precision.parse("train_bf16attn")
    assert (b.score, b.pv, b.conv, b.wgx, b.wgy, b.dxw, b.sbw) == (PREC_BF16, PREC_BF16, PREC_F16X3, PREC_F16, PREC_F16, PREC_F16, None)
    f = Precision.parse("train_f16x3")
    assert (f.wgx, f.wgy, f.dxw, f.sbw) == (None,) * 4 and Precision.parse("fp32").wgx is None
    assert Precision.parse("proj=f16x3,conv=f16x3,wgx=fp16").wgy is None
    assert training_precision(m, loss_scaled=True) is m
    p = training_precision(m, loss_scaled=False)
    assert p.pv == PREC_F16X3 and (p.wgx, p.wgy, p.dxw, p.sbw) == (None,) * 4 and p.conv == PREC_F16X3
    pb = training_precision(b, loss_scaled=False)
    assert pb.score == PREC_BF16 and (pb.wgx, pb.wgy, pb.dxw) == (None,) * 3
    assert "wgx=fp16" in repr(m) and "wgx=layer" in repr(f)
