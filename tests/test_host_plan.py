"""Host-side budget of a multi-rank run (helen_amd/host_plan.py) and the rank supervisor of predict_gpu
(helen_amd.predict.run_ranks: one failing rank takes its siblings down, models/predict_gpu.py:223 `join=True`)."""
import os
import sys
import time

import pytest

from helen_amd import host_plan
from helen_amd.host_plan import plan_host

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_parse_cpulist():
    assert host_plan.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert host_plan.parse_cpulist("") == []


def test_readers_are_capped_by_the_usable_cpus():
    # the GPU box of this project: 16 CPUs' worth of quota, 8 readers asked for, one rank
    p = plan_host([0], 8, 4096, usable=16, allowed=list(range(256)), shm_free=64 << 30, local_cpus=lambda d: (None, None))
    assert [r.reader_workers for r in p.ranks] == [8] and p.ranks[0].slots == 5
    # eight ranks on the same quota: (16 - 2 x 8) // 8 = 0 -> one reader each, and the plan says the host is the bound
    p = plan_host(list(range(8)), 8, 4096, usable=16, allowed=list(range(256)), shm_free=64 << 30,
                  local_cpus=lambda d: (None, None))
    assert [r.reader_workers for r in p.ranks] == [1] * 8
    d = p.as_dict()
    assert d["predicted_bound"] == "host readers" and d["predicted_host_ceiling_windows_per_s"] < d["predicted_device_ceiling_windows_per_s"]
    assert any("GRANTED" in n for n in p.notes)
    # a full node: 128 usable CPUs, 8 ranks -> (128 - 16) // 8 = 14 >= 8: the request stands, device-bound
    p = plan_host(list(range(8)), 8, 4096, usable=128, allowed=list(range(128)), shm_free=256 << 30,
                  local_cpus=lambda d: (None, None))
    assert [r.reader_workers for r in p.ranks] == [8] * 8 and p.as_dict()["predicted_bound"] == "device"
    # -w 0 (inline reading) stays 0
    p = plan_host([0, 1], 0, 4096, usable=4, allowed=[0, 1, 2, 3], shm_free=64 << 30, local_cpus=lambda d: (None, None))
    assert [r.reader_workers for r in p.ranks] == [0, 0]


def test_uncapped_request(monkeypatch):
    monkeypatch.setenv("HELEN_READERS_UNCAPPED", "1")
    p = plan_host([0, 1], 8, 4096, usable=4, allowed=[0, 1, 2, 3], shm_free=64 << 30, local_cpus=lambda d: (None, None))
    assert [r.reader_workers for r in p.ranks] == [8, 8]


def test_numa_pinning():
    # two sockets: GPUs 0-3 on node 0 (CPUs 0-63), GPUs 4-7 on node 1 (CPUs 64-127)
    local = lambda d: (d // 4, list(range(64 * (d // 4), 64 * (d // 4) + 64)))        # noqa: E731
    p = plan_host(list(range(8)), 6, 4096, usable=128, allowed=list(range(128)), shm_free=256 << 30, local_cpus=local)
    assert [r.numa_node for r in p.ranks] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert p.ranks[0].cpus == list(range(64)) and p.ranks[7].cpus == list(range(64, 128))
    # the allowed mask is honoured (a container that may only use CPUs 0-15 and 64-79)
    allowed = list(range(16)) + list(range(64, 80))
    p = plan_host([0, 4], 6, 4096, usable=32, allowed=allowed, shm_free=256 << 30, local_cpus=local)
    assert p.ranks[0].cpus == list(range(16)) and p.ranks[1].cpus == list(range(64, 80))
    # a node whose allowed share is smaller than what its ranks run is not pinned
    p = plan_host(list(range(8)), 6, 4096, usable=128, allowed=list(range(8)) + list(range(64, 128)), shm_free=256 << 30,
                  local_cpus=local)
    assert p.ranks[0].cpus is None and p.ranks[4].cpus == list(range(64, 128))
    assert any("NOT PINNED" in n for n in p.notes)
    # one node for every device, or no information: no pinning
    p = plan_host([0, 1], 4, 4096, usable=32, allowed=list(range(32)), shm_free=64 << 30,
                  local_cpus=lambda d: (0, list(range(32))))
    assert all(r.cpus is None for r in p.ranks)
    p = plan_host([0, 1], 4, 4096, usable=32, allowed=list(range(32)), shm_free=64 << 30, local_cpus=lambda d: (None, None))
    assert all(r.cpus is None for r in p.ranks)


def test_pinning_can_be_turned_off(monkeypatch):
    monkeypatch.setenv("HELEN_PIN", "0")
    local = lambda d: (d, list(range(8 * d, 8 * d + 8)))                                  # noqa: E731
    p = plan_host([0, 1], 2, 4096, usable=16, allowed=list(range(16)), shm_free=64 << 30, local_cpus=local)
    assert all(r.cpus is None for r in p.ranks)


def test_slot_budget_over_all_ranks():
    per_slot = 4096 * host_plan.SLOT_BYTES_PER_WINDOW
    p = plan_host(list(range(8)), 4, 4096, usable=128, allowed=list(range(128)), shm_free=int(8 * 5 * per_slot * 1.3),
                  local_cpus=lambda d: (None, None))
    assert p.ranks[0].slots == 5 and p.shm_need == 8 * 5 * per_slot
    p = plan_host(list(range(8)), 4, 4096, usable=128, allowed=list(range(128)), shm_free=int(8 * 4 * per_slot),
                  local_cpus=lambda d: (None, None))
    assert p.ranks[0].slots == 3 and any("THREE PER RANK" in n for n in p.notes)
    p = plan_host(list(range(8)), 4, 4096, usable=128, allowed=list(range(128)), shm_free=per_slot,
                  local_cpus=lambda d: (None, None))
    assert p.ranks[0].slots == 3 and any("TEMP DIRECTORY" in n for n in p.notes)
    # a short run needs no more slots than it has device calls
    p = plan_host([0], 4, 4096, calls_per_rank=2, usable=16, allowed=list(range(16)), shm_free=64 << 30,
                  local_cpus=lambda d: (None, None))
    assert p.ranks[0].slots == 2
    # every rank names its slots after the parent and itself: what a killed rank leaves behind can be swept
    assert p.ranks[0].slot_prefix.startswith("helen_slot_%d_0_" % os.getpid())


def test_sweep_slots(tmp_path):
    for name in ("helen_slot_77_0_abc", "helen_slot_77_1_def", "helen_slot_78_0_xyz", "other"):
        (tmp_path / name).write_bytes(b"x")
    assert host_plan.sweep_slots(["helen_slot_77_0_", "helen_slot_77_1_"], directories=(str(tmp_path),)) == 2
    assert sorted(os.listdir(tmp_path)) == ["helen_slot_78_0_xyz", "other"]


def test_a_failing_rank_takes_its_siblings_down(tmp_path):
    """models/predict_gpu.py:223: mp.spawn(join=True) ends every process as soon as one fails.  Rank 1 exits with 3
    while ranks 0 and 2 would run for a minute: run_ranks returns within seconds, the siblings were unwound through
    their own tear-down (SIGTERM -> SystemExit -> finally)."""
    import rank_targets
    from helen_amd.predict import run_ranks
    d = str(tmp_path)
    t0 = time.time()
    # ranks 0 and 2 sleep, rank 1 fails once both are up
    results, failed = _run_mixed(run_ranks, [rank_targets.sleeper, rank_targets.failing, rank_targets.sleeper],
                                 [(0, d), (1, d), (2, d)])
    took = time.time() - t0
    assert failed[0] == (1, 3)
    assert took < 30, took
    assert os.path.exists(os.path.join(d, "unwound_0")) and os.path.exists(os.path.join(d, "unwound_2"))
    assert 0 not in results and 2 not in results


def _run_mixed(run_ranks, targets, argsets, **kw):
    """run_ranks takes one target: dispatch on the rank."""
    return run_ranks(_dispatch, [(a[0], a[1], [t.__name__ for t in targets]) for a in argsets], **kw)


def _dispatch(rank, marker_dir, names, result_q):
    import rank_targets
    if names[rank] == "failing":
        rank_targets.failing(rank, marker_dir, result_q,
                             wait_for=[r for r, n in enumerate(names) if n in ("sleeper", "stubborn")])
    else:
        getattr(rank_targets, names[rank])(rank, marker_dir, result_q)


def test_a_rank_that_ignores_sigterm_is_killed(tmp_path):
    import rank_targets
    from helen_amd.predict import run_ranks
    d = str(tmp_path)
    t0 = time.time()
    results, failed = _run_mixed(run_ranks, [rank_targets.stubborn, rank_targets.failing], [(0, d), (1, d)],
                                 grace_seconds=1.0)
    assert failed[0] == (1, 3) and time.time() - t0 < 30


def test_all_ranks_fine(tmp_path):
    import rank_targets
    from helen_amd.predict import run_ranks
    results, failed = _run_mixed(run_ranks, [rank_targets.quick] * 3, [(r, str(tmp_path)) for r in range(3)])
    assert failed == [] and sorted(results) == [0, 1, 2] and results[2]["rank"] == 2


@pytest.mark.parametrize("signum", [2, 15])
def test_an_interrupted_parent_takes_its_ranks_down(tmp_path, signum):
    """The ranks live in process groups of their own, so Ctrl-C / a scheduler's SIGTERM reaches only the parent: run_ranks
    passes it on (SIGTERM -> the ranks' own tear-down, the grace period, the group kill) and re-raises, instead of leaving
    the ranks on the GPUs while the parent hangs in multiprocessing's exit handler."""
    import signal
    import subprocess
    import sys
    d = str(tmp_path)
    tests = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PYTHONPATH=tests + os.pathsep + os.path.dirname(tests) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    parent = subprocess.Popen([sys.executable, os.path.join(tests, "rank_targets.py"), d], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 60
    while time.time() < deadline and not all(os.path.exists(os.path.join(d, "started_%d" % r)) for r in (0, 1)):
        time.sleep(0.05)
    assert os.path.exists(os.path.join(d, "started_1")), "the ranks did not start"
    t0 = time.time()
    parent.send_signal(signum)
    out, err = parent.communicate(timeout=40)
    took = time.time() - t0
    assert took < 20, (took, err)
    assert parent.returncode != 0
    assert "INTERRUPTED" in err, err
    assert os.path.exists(os.path.join(d, "unwound_0")) and os.path.exists(os.path.join(d, "unwound_1")), err
