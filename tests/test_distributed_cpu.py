"""The N > 1 path on CPU: two processes over gloo.  Ranks take disjoint round-robin shards, never
exchange data, and only barrier + max-reduce the elapsed time (bench.py's rule)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
from helen_amd import dist_util
from helen_amd.file_manager import shard_round_robin
rank, local_rank, world = dist_util.env_world()
dist = dist_util.init_distributed("gloo")
assert dist is not None and dist.get_world_size() == 2
files = ["img_%%02d.h5" %% i for i in range(7)]
mine = shard_round_robin(files, world)[rank]
assert mine == dist_util.shard_for_rank(files, rank, world)
dist_util.barrier(dist)
t = dist_util.max_over_ranks(dist, 1.0 + rank)          # rank 1 is "slower"
n = dist_util.sum_over_ranks(dist, len(mine))
import torch
gathered = [None, None]
dist.all_gather_object(gathered, mine)                   # test-only: check the shards
flat = sorted(sum(gathered, []))
assert flat == files, flat                               # disjoint and complete
if rank == 0:
    print(json.dumps({"max_time": t, "total": n, "shard0": mine}))
dist_util.barrier(dist)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_and_timing_rule(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["max_time"] == 2.0 and r["total"] == 7.0
    assert r["shard0"] == ["img_00.h5", "img_02.h5", "img_04.h5", "img_06.h5"]


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` launches N ranks itself -- and says so loudly, with a non-zero exit code, when
    fewer than N GPUs are visible (none in the build container) instead of quietly running one rank."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2
    assert "--gpus 2 but only 0 GPU(s) visible" in out.stderr
    # a rank count that disagrees with the launcher's WORLD_SIZE is refused as well
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE is 4" in out.stderr


def test_bench_plans_eight_ranks_without_a_device():
    """The driver's 8-GPU command shape on a machine without the GPUs: `bench.py --gpus 8 --steps 20 --warmup 5` under a
    launcher's environment (WORLD_SIZE=8), dry (`--plan-only`): the arguments parse, every default leg is planned, the
    end-to-end leg is sized against /dev/shm for ALL eight ranks (and says when it shrinks), the host plan covers eight
    ranks within the usable CPUs -- and no device is touched (there is none here).  Ranks other than 0 print nothing."""
    import json
    env = dict(os.environ, WORLD_SIZE="8", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--plan-only"]
    out = subprocess.run(base, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["plan_only"] and plan["n_gpus"] == 8 and plan["windows_per_step"] == 4096
    assert plan["legs"] == {"cpu_baseline": False, "host_path": True, "modes": False, "margins": True, "end_to_end": True}      # (cpu_baseline and modes: one rank only)
    e = plan["end_to_end"]
    assert e["image_files"] == 128 and e["contigs"] == 256 and 8192 <= e["windows_per_rank"] <= 300000
    assert e["shrunk"] == (e["windows_per_rank"] != 300000)
    assert e["directory"] in ("/dev/shm", None) and (e["directory"] is None or e["ram_bytes_needed"] * 1.1 < e["shm_free_bytes"])
    hp = e["host_plan"]
    assert hp["n_ranks"] == 8 and len(hp["ranks"]) == 8 and all(r["reader_workers"] >= 1 for r in hp["ranks"])
    assert sum(hp["reader_workers_per_rank"]) + 2 * 8 <= max(hp["usable_cpus"], 3 * 8)
    # another rank of the same launch says nothing; a WORLD_SIZE that contradicts --gpus is refused
    quiet = subprocess.run(base, capture_output=True, text=True, timeout=300, env=dict(env, RANK="5", LOCAL_RANK="5"))
    assert quiet.returncode == 0 and quiet.stdout.strip() == ""
    wrong = subprocess.run(base, capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="4"))
    assert wrong.returncode == 2 and "WORLD_SIZE is 4" in wrong.stderr
    # started plainly, it answers for the ranks it would become
    plain = subprocess.run(base, capture_output=True, text=True, timeout=300,
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert plain.returncode == 0 and json.loads(plain.stdout.strip().splitlines()[-1])["n_gpus"] == 8
