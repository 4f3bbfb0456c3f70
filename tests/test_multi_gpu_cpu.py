"""World-size-2 (gloo, CPU) test of the N>1 path of bench.py: pairs are sharded by batch across ranks with no
data-path collective; each rank times its own steps between two barriers and the job reports
global pairs / max-over-ranks time (SURVEY.md §8(e), weak scaling).  The GPU forward is replaced by a stub
step here — the distributed bookkeeping is what is under test."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from craft_amd.dist import shard_batch, timed_steps, aggregate_throughput
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    # global batch of 8 pairs -> contiguous shards, every pair exactly once
    idx = shard_batch(8, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, idx)
    assert sorted(sum(gathered, [])) == list(range(8)), gathered
    # rank 1 is made slower: the job-level time must be the max over ranks
    def step():
        time.sleep(0.02 if rank == 0 else 0.05)
    dt = timed_steps(step, steps=3, warmup=1, sync=lambda: None)
    value, dt_max = aggregate_throughput(pairs_per_rank_step=len(idx), steps=3, dt=dt)
    if rank == 0:
        print(json.dumps({"value": value, "dt_max": dt_max, "dt0": dt}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_sharding_and_max_time(tmp_path):
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
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dt_max"] >= 0.14            # 3 steps of the slow rank (0.05 s)
    assert out["dt0"] < out["dt_max"] + 1e-6 or out["dt0"] >= 0.14   # rank 0 waited at the closing barrier
    assert abs(out["value"] - 8 * 3 / out["dt_max"]) < 1e-6


def test_shard_batch_is_a_partition():
    from craft_amd.dist import shard_batch
    for n, w in ((8, 1), (8, 2), (10, 4), (3, 8)):
        parts = [shard_batch(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


WORKER8 = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from craft_amd.dist import shard_batch, timed_steps, aggregate_throughput
    from craft_amd.train import FlatAdamW
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 8
    # timing protocol at the driver's largest N: rank 5 is the slow one
    def step():
        time.sleep(0.06 if rank == 5 else 0.01)
    dt = timed_steps(step, steps=2, warmup=1, sync=lambda: None)
    value, dt_max = aggregate_throughput(pairs_per_rank_step=4, steps=2, dt=dt)
    # the training exchange: ONE all-reduce of the flat gradient; 8 different per-rank gradients -> their mean after the 1/world factor
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
    opt = FlatAdamW(net.parameters())
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(opt.numel, generator=g)
    opt.flat_grad.copy_(mine)
    mul = opt.allreduce_grads()
    want = sum(torch.randn(opt.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    assert mul == 1.0 / world
    assert torch.allclose(opt.flat_grad * mul, want, rtol=1e-6, atol=1e-6)
    # replica broadcast (Trainer.sync_replicas): rank 3's weights everywhere
    opt.flat.fill_(float(rank))
    opt.step_count = rank
    opt.broadcast_state(src=3)
    assert float(opt.flat.min()) == float(opt.flat.max()) == 3.0 and opt.step_count == 3
    if rank == 0:
        print(json.dumps({"value": value, "dt_max": dt_max}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_eight_rank_timing_allreduce_and_broadcast(tmp_path):
    """The driver's N = 8 launch on CPU (gloo): barrier / max-over-ranks timing, whole-job pairs / slowest rank's time, the flat
    gradient all-reduce as a mean over 8 different gradients, the replica broadcast from a non-zero source rank."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dt_max"] >= 0.12                          # 2 steps of the slow rank
    assert abs(out["value"] - 8 * 4 * 2 / out["dt_max"]) < 1e-6


def test_bench_graph_probe_child_runs_outside_the_process_group(monkeypatch):
    """bench.py tries the hipGraph capture in a CHILD first (a crash inside the runtime must not cost the line): the child must not inherit
    the rank / rendezvous variables of a torch.distributed.run launch (it would try to join the parent's group) and runs on this rank's device;
    only `graph-probe ok` with exit code 0 counts."""
    import argparse
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_for_probe", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, seen.get("rc", 0), stdout=seen.get("out", "graph-probe ok\n"), stderr="boom\n")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    for k, v in (("RANK", "3"), ("LOCAL_RANK", "3"), ("WORLD_SIZE", "8"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500"),
                 ("CRAFT_FORCE_COLLECTIVES", "1"), ("CRAFT_BENCH_BACKEND", "gloo")):
        monkeypatch.setenv(k, v)
    a = argparse.Namespace(batch=4, height=448, width=1024, iters=12, precision="mixed")
    ok, note = bench.graph_probe_child(a, 3)
    assert ok and note == ""
    env, cmd = seen["env"], seen["cmd"]
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRAFT_FORCE_COLLECTIVES"} & set(env)
    assert env["CRAFT_PROBE_DEVICE"] == "3" and env["CRAFT_BENCH_BACKEND"] == "gloo"
    assert "--graph-probe" in cmd and cmd[cmd.index("--batch") + 1] == "4" and cmd[cmd.index("--width") + 1] == "1024"
    seen["rc"] = -11                                # the child died of SIGSEGV
    ok, note = bench.graph_probe_child(a, 0)
    assert not ok and "-11" in note
    seen["rc"], seen["out"] = 0, ""                 # exit 0 without the marker (e.g. an early return) is not a pass either
    assert not bench.graph_probe_child(a, 0)[0]
