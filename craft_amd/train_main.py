"""Training driver: the reference's ``train.py`` / ``train_ddp.py`` main loop (train.py:176-262, train_ddp.py:187-262) over the pieces of this
package -- ``train_data`` (file decode + GPU augmentation, asynchronous feed), ``Trainer.step`` (forward, sequence loss, backward, one RCCL
all-reduce, clip, fused AdamW, OneCycle), reference-layout checkpoints and the evaluation harness for the periodic validation.

    python -m craft_amd.train_main --name craft --stage chairs --validation chairs --output checkpoints --num_steps 120000 \\
        --lr 0.00025 --image_size 368 496 --wdecay 0.0001 --batch_size 8 --craft --f2 full --setrans            # train-craft-f2full.sh
    python -m torch.distributed.run --nproc-per-node 8 -m craft_amd.train_main ...                                # one rank per GPU

The flags keep the reference's names and defaults (train.py:313-404); ``--batch_size`` is PER RANK as in train_ddp.py.  ``--mixed_precision``
selects the reference's fp16-AMP arithmetic (policy ``train_amp_fp16``); without it the library default ``mixed`` runs (f16x3 operands, fp16
attention products under the loss scale); ``--hip_precision`` names any other policy.  Dataset roots default to the reference's
``datasets/<name>`` layout; only the walkers of ``flow_datasets`` exist (FlyingChairs, MPI-Sintel, KITTI): the sintel stage mixes
100 x clean + 100 x final (+ 200 x KITTI when its root exists) -- FlyingThings3D / HD1K shares of datasets.py:548 are skipped with a note.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
from typing import List, Optional

import torch


def parse(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="Train CRAFT on the HIP path (the reference's train.py / train_ddp.py command line)")
    ap.add_argument("--name", default="craft")
    ap.add_argument("--stage", required=True, choices=["chairs", "sintel", "kitti"])
    ap.add_argument("--validation", nargs="+", default=[])
    ap.add_argument("--restore_ckpt")
    ap.add_argument("--loadopt", dest="load_optimizer_state", action="store_true")
    ap.add_argument("--loadsched", dest="load_scheduler_state", action="store_true")
    ap.add_argument("--trust-checkpoint", dest="trust_checkpoint", action="store_true")
    ap.add_argument("--output", default="checkpoints")
    ap.add_argument("--lr", type=float, default=0.00002)
    ap.add_argument("--num_steps", type=int, default=100000)
    ap.add_argument("--batch_size", type=int, default=6)
    ap.add_argument("--workers", dest="num_workers", type=int, default=4)
    ap.add_argument("--image_size", type=int, nargs=2, default=[384, 512])
    ap.add_argument("--mixed_precision", action="store_true")
    ap.add_argument("--hip_precision", default=None)
    ap.add_argument("--wdecay", type=float, default=0.00005)
    ap.add_argument("--epsilon", type=float, default=1e-8)
    ap.add_argument("--clip", type=float, default=1.0)
    ap.add_argument("--gamma", type=float, default=0.8)
    ap.add_argument("--add_noise", action="store_true")
    ap.add_argument("--shiftprob", dest="shift_aug_prob", type=float, default=0.0)
    ap.add_argument("--shiftsigmas", dest="shift_sigmas", default="16,10")
    ap.add_argument("--freeze_bn", action="store_true")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--val_freq", type=int, default=10000)
    ap.add_argument("--print_freq", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--chairs_root", default="datasets/FlyingChairs_release/data")
    ap.add_argument("--sintel_root", default="datasets/Sintel")
    ap.add_argument("--kitti_root", default="datasets/KITTI")
    # model switches (train.py:316-404; everything else keeps default_args' value)
    ap.add_argument("--craft", action="store_true")
    ap.add_argument("--setrans", dest="use_setrans", action="store_true")
    ap.add_argument("--radius", dest="corr_radius", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--f1", dest="f1trans", default="none", choices=["none", "shared", "private"])
    ap.add_argument("--f2", dest="f2trans", default="full", choices=["none", "full"])
    ap.add_argument("--position_only", action="store_true")
    ap.add_argument("--position_and_content", action="store_true")
    ap.add_argument("--num_heads", type=int, default=1)
    ap.add_argument("--posr", dest="pos_bias_radius", type=int, default=7)
    ap.add_argument("--f2posw", dest="f2_pos_code_weight", type=float, default=0.5)
    ap.add_argument("--f2radius", dest="f2_attn_mask_radius", type=int, default=-1)
    ap.add_argument("--intermodes", dest="inter_num_modes", type=int, default=4)
    ap.add_argument("--intramodes", dest="intra_num_modes", type=int, default=4)
    ap.add_argument("--f2modes", dest="f2_num_modes", type=int, default=4)
    ap.add_argument("--interqknobias", dest="inter_qk_have_bias", action="store_false")
    ap.add_argument("--interpos", dest="inter_pos_code_type", default="bias", choices=["lsinu", "bias"])
    ap.add_argument("--interposw", dest="inter_pos_code_weight", type=float, default=0.5)
    ap.add_argument("--intrapos", dest="intra_pos_code_type", default="bias", choices=["lsinu", "bias"])
    ap.add_argument("--intraposw", dest="intra_pos_code_weight", type=float, default=1.0)
    # accepted for command-line compatibility: the reference's DataParallel device list (here: one process per GPU under
    # torch.distributed.run) and --model_name; the model-family switches this package does not build (--nogma, --raft,
    # --upsample-learn) are NOT accepted
    ap.add_argument("--gpus", type=int, nargs="+", default=None)
    ap.add_argument("--model_name", default="")
    ns = ap.parse_args(argv)
    if ns.gpus and len(ns.gpus) > 1 and int(os.environ.get("WORLD_SIZE", 1)) == 1:
        print(f"[train_main] --gpus {ns.gpus}: this driver runs one process per GPU; start it under `python -m torch.distributed.run "
              f"--nproc-per-node {len(ns.gpus)} -m craft_amd.train_main ...` for {len(ns.gpus)} GPUs (continuing on one)", flush=True)
    return ns


def fetch_sources(ns: argparse.Namespace) -> List:
    """fetch_dataloader's dataset mix of the stage (datasets.py:509-567) over the walkers that exist here."""
    from .flow_datasets import KITTI, FlyingChairs, MpiSintel
    from .train_data import TrainSource, make_augmentor
    crop = tuple(ns.image_size)
    sig = tuple(int(v) for v in str(ns.shift_sigmas).split(","))
    mk = lambda ds, key, rep=1: TrainSource(ds, make_augmentor(ds, key, crop, ns.shift_aug_prob, sig), repeat=rep)      # noqa: E731
    if ns.stage == "chairs":
        return [mk(FlyingChairs("training", ns.chairs_root), "chairs")]
    if ns.stage == "kitti":
        return [mk(KITTI("training", ns.kitti_root), "kitti")]
    out = [mk(MpiSintel("training", ns.sintel_root, "clean"), "sintel", 100), mk(MpiSintel("training", ns.sintel_root, "final"), "sintel", 100)]
    if os.path.isdir(os.path.join(ns.kitti_root, "training")):
        out.append(mk(KITTI("training", ns.kitti_root), "sintel/kitti", 200))
    print("[train_main] sintel stage: FlyingThings3D / HD1K shares of datasets.py:548 are not available here (no walker): "
          f"mixing {len(out)} dataset(s)", flush=True)
    return out


def validate(model, ns: argparse.Namespace, dev) -> dict:
    """train.py:265-293: the harness' validators on the training model in eval mode."""
    from . import evaluate
    results = {}
    was_training = model.training
    model.eval()
    for name in ns.validation:
        if name == "chairs":
            results.update(evaluate.validate_chairs(model, ns.chairs_root, ns.iters, device=dev))
        elif name == "sintel":
            results.update(evaluate.validate_sintel(model, ns.sintel_root, ns.iters, device=dev))
        elif name == "kitti":
            results.update(evaluate.validate_kitti(model, ns.kitti_root, ns.iters, device=dev))
        else:
            print(f"[train_main] validation set {name!r} has no walker here: skipped", flush=True)
    results = {k: float(v) for k, v in results.items() if isinstance(v, (int, float))}
    if was_training:
        model.train()
        if ns.freeze_bn and ns.stage != "chairs":
            model.freeze_bn()
    return results


def main(argv=None) -> Optional[str]:
    ns = parse(argv)
    if not torch.cuda.is_available():
        raise SystemExit("craft_amd.train_main needs a GPU (the hot path has no CPU fallback)")
    from . import CRAFT, default_args
    from .train import Trainer, load_checkpoint, save_checkpoint
    from .train_data import seed_workers, train_batches_async
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("CRAFT_BENCH_BACKEND", "nccl"), init_method="env://")
    # from-scratch initial weights follow --seed (the reference seeds torch / numpy before it builds the model, train.py:407-408);
    # the per-rank augmentation draws are re-seeded below
    import random
    torch.manual_seed(ns.seed)
    np.random.seed(ns.seed)
    random.seed(ns.seed)
    policy = ns.hip_precision or ("train_amp_fp16" if ns.mixed_precision else "mixed")
    known = vars(default_args())
    over = {k: v for k, v in vars(ns).items() if k in known}
    over["hip_precision"] = policy
    model = CRAFT(default_args(**over)).to(dev)
    freeze = ns.freeze_bn and ns.stage != "chairs"                    # train.py:196-197
    tr = Trainer(model, lr=ns.lr, wdecay=ns.wdecay, epsilon=ns.epsilon, num_steps=ns.num_steps, clip=ns.clip, gamma=ns.gamma, iters=ns.iters,
                 add_noise=ns.add_noise, freeze_bn=freeze)
    log = {"total_steps": 0, "val_steps": [], "val_results": {}}
    if ns.restore_ckpt:
        msg, saved = load_checkpoint(ns.restore_ckpt, model, tr.optimizer, tr.scheduler, ns.load_optimizer_state, ns.load_scheduler_state,
                                     trusted=ns.trust_checkpoint)
        if saved and ns.load_scheduler_state:
            log.update(saved)
            tr.total_steps = int(log.get("total_steps", 0))
        tr.sync_replicas()
        if rank == 0:
            print(f"[train_main] restored {ns.restore_ckpt}: {msg}", flush=True)
    if rank == 0:
        os.makedirs(ns.output, exist_ok=True)
        print(f"[train_main] {sum(p.numel() for p in model.parameters())} parameters, policy {policy}, {world} rank(s) x batch {ns.batch_size}", flush=True)
    seed_workers(ns.seed + rank)
    sources = fetch_sources(ns)
    if rank == 0:
        print(f"[train_main] training with {sum(len(s) for s in sources)} image pairs", flush=True)
    feed = train_batches_async(sources, ns.batch_size, dev, seed=ns.seed, rank=rank, world=world, workers=max(1, ns.num_workers))
    run, t0 = {}, time.time()
    path = None
    try:
        for im1, im2, flow, valid in feed:
            m = tr.step(im1, im2, flow, valid)
            log["total_steps"] = tr.total_steps
            for k, v in m.items():
                run[k] = run.get(k, 0.0) + float(v)
            if tr.total_steps % ns.print_freq == 0 and rank == 0:      # the reference's Logger: running means over print_freq steps
                dt = (time.time() - t0) / ns.print_freq
                print(f"[{tr.total_steps:6d}, lr {tr.scheduler.get_last_lr()[0]:.7f}] " + ", ".join(f"{k} {v / ns.print_freq:.4f}" for k, v in run.items())
                      + f", {dt * 1e3:.1f} ms/step", flush=True)
                run, t0 = {}, time.time()
            if tr.total_steps % ns.val_freq == 0 or tr.total_steps >= ns.num_steps:
                tr.sync_buffers()                                      # DDP's broadcast_buffers: rank 0's BatchNorm statistics everywhere
                if rank == 0:                                          # train.py:238-243: checkpoint, then validate
                    save_checkpoint(os.path.join(ns.output, f"{tr.total_steps}_{ns.name}.pth"), model, tr.optimizer, tr.scheduler, log)
                    if ns.validation:
                        res = validate(model, ns, dev)
                        log["val_steps"].append(tr.total_steps)
                        for k, v in res.items():
                            log["val_results"].setdefault(k, []).append(float(v))
                        print(f"[train_main] step {tr.total_steps}: " + ", ".join(f"{k} {float(v):.4f}" for k, v in res.items()), flush=True)
                if world > 1:
                    torch.distributed.barrier()
            if tr.total_steps >= ns.num_steps:
                break
        if rank == 0:                                                  # train.py:205-207: the final checkpoint (with the last validation logged)
            path = os.path.join(ns.output, f"{ns.name}.pth")
            save_checkpoint(path, model, tr.optimizer, tr.scheduler, log)
    finally:
        feed.close()
        if world > 1:
            torch.distributed.destroy_process_group()
    return path


if __name__ == "__main__":
    main()
