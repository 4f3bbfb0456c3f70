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
