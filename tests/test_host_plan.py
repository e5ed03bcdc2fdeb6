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


def test_the_plan_prices_readers_with_the_storage_it_finds(tmp_path):
    """host_plan.READER_RATE: one reader's windows/s per storage class (profiles/r04_reader_variants.txt).  A directory of
    deflated images is host-bound where a contiguous one is device-bound, and the plan says so before anything starts."""
    import numpy as np

    from helen_amd import native_io
    from helen_amd.host_plan import READER_RATE, reader_rate, storage_of_files
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    if not native_io.available():
        pytest.skip("libhelen_io.so not built")
    img = make_images(3, seed=1)
    files = []
    for name, kw in (("a_plain", {}), ("b_chunked", dict(chunks=(100, 90))), ("c_gzip", dict(gzip=4)),
                     ("d_latest", dict(libver="latest")), ("e_paged", dict(libver="latest", chunks=(1, 45)))):
        files.append(str(tmp_path / (name + ".h5")))
        write_image_file(files[-1], img, **kw)
    assert [native_io.image_storage(f) for f in files] == ["contiguous", "chunked", "deflate", "contiguous", "libhdf5"]
    assert storage_of_files(files) == {"contiguous": 2, "chunked": 1, "deflate": 1, "libhdf5": 1}
    assert reader_rate({"contiguous": 5}) == READER_RATE["contiguous"]
    mixed = reader_rate({"contiguous": 1, "deflate": 1})
    assert READER_RATE["deflate"] < mixed < 2 * READER_RATE["deflate"]        # harmonic: the slow half dominates
    fast = plan_host([0], 8, 4096, usable=16, allowed=list(range(16)), shm_free=1 << 40, storage=[{"contiguous": 16}])
    # (four readers: with libdeflate eight of them inflate 100 k windows/s, more than the device takes)
    slow = plan_host([0], 4, 4096, usable=16, allowed=list(range(16)), shm_free=1 << 40, storage=[{"deflate": 16}])
    assert fast.as_dict()["predicted_bound"] == "device" and slow.as_dict()["predicted_bound"] == "host readers"
    assert slow.as_dict()["reader_windows_per_s_each"] == [round(READER_RATE["deflate"])]
    import io
    text = io.StringIO()
    slow.describe(out=text)
    assert "HOST-BOUND" in text.getvalue() and "DEFLATE" in text.getvalue()
    text = io.StringIO()
    fast.describe(out=text)
    assert "HOST-BOUND" not in text.getvalue() and "CONTIGUOUS" in text.getvalue()
    # check_images reports the class and the predicted rate per file
    from helen_amd.check_images import image_directory_report
    rep = image_directory_report(str(tmp_path), strict=True)
    got = {os.path.basename(e["path"]): (e["storage_class"], e["reader_path"]) for e in rep["files"]}
    assert got["c_gzip.h5"] == ("deflate", "direct scanner") and got["e_paged.h5"] == ("libhdf5", "libhdf5")
    assert all(e["predicted_windows_per_s_per_reader"] == READER_RATE[e["storage_class"]] for e in rep["files"])


def _fake_node(tmp_path, gpus=8, cpus_per_socket=96):
    """A 2-socket, 8-GPU box as sysfs shows it: KFD topology nodes 0-1 = the sockets (no SIMDs), 2-9 = the GPUs (four per
    socket, PCI buses 0x05.. / 0x85..), and /sys/bus/pci/devices/<address>/{numa_node, local_cpulist}."""
    topo, pci = tmp_path / "kfd_nodes", tmp_path / "pci"
    for n in range(2):
        (topo / str(n)).mkdir(parents=True)
        (topo / str(n) / "properties").write_text("cpu_cores_count %d\nsimd_count 0\nlocation_id 0\ndomain 0\n" % cpus_per_socket)
    for g in range(gpus):
        socket = g // (gpus // 2)
        bus = (0x05 if socket == 0 else 0x85) + 0x10 * (g % (gpus // 2))
        (topo / str(2 + g)).mkdir(parents=True)
        (topo / str(2 + g) / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        d = pci / ("0000:%02x:00.0" % bus)
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % socket)
        lo = socket * cpus_per_socket
        (d / "local_cpulist").write_text("%d-%d\n" % (lo, lo + cpus_per_socket - 1))
    return str(topo), str(pci)


def test_eight_ranks_on_a_two_socket_node(tmp_path, monkeypatch):
    """BASELINE.json configs[2] without the node: plan_host on a fake sysfs tree of 2 sockets x 96 CPUs and 8 GPUs (the
    PCI address of a device comes from the KFD topology, its NUMA node and CPUs from /sys/bus/pci -- no torch, no device
    context) -- with all 192 CPUs, and under a 16-CPU cgroup quota."""
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "HELEN_PIN", "HELEN_READERS_UNCAPPED"):
        monkeypatch.delenv(k, raising=False)
    topo, pci = _fake_node(tmp_path)
    addrs = host_plan.kfd_gpu_addresses(topo)
    assert addrs == ["0000:05:00.0", "0000:15:00.0", "0000:25:00.0", "0000:35:00.0",
                     "0000:85:00.0", "0000:95:00.0", "0000:a5:00.0", "0000:b5:00.0"]
    assert host_plan.device_pci_address(5, topo) == "0000:95:00.0" and host_plan.device_pci_address(8, topo) is None
    assert host_plan.visible_gpu_addresses(topo, {"HIP_VISIBLE_DEVICES": "6,1"}) == ["0000:a5:00.0", "0000:15:00.0"]
    assert host_plan.visible_gpu_addresses(topo, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1"}) == ["0000:95:00.0"]
    assert host_plan.visible_gpu_addresses(topo, {"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) == []
    assert host_plan.device_local_cpus("0000:95:00.0", pci) == (1, list(range(96, 192)))

    def local(d):
        return host_plan.device_local_cpus(host_plan.device_pci_address(d, topo), pci)
    storage = [{"contiguous": 16}] * 8
    # the whole machine: every rank gets the 8 readers it asked for, pinned to its GPU's socket, device-bound
    full = plan_host(list(range(8)), 8, 4096, usable=192, allowed=list(range(192)), shm_free=1 << 40, local_cpus=local,
                     storage=storage)
    d = full.as_dict()
    assert d["reader_workers_per_rank"] == [8] * 8 and d["predicted_bound"] == "device"
    assert [r.numa_node for r in full.ranks] == [0] * 4 + [1] * 4
    assert full.ranks[0].cpus == list(range(96)) and full.ranks[7].cpus == list(range(96, 192))
    assert d["predicted_device_ceiling_windows_per_s"] == 8 * host_plan.DEVICE_WINDOWS_PER_S
    # a 16-CPU grant on the same box: (16 - 2 x 8) // 8 = 0 -> one reader per rank, host-bound, and the plan says so
    small = plan_host(list(range(8)), 8, 4096, usable=16, allowed=list(range(16)), shm_free=1 << 40, local_cpus=local,
                      storage=storage)
    d = small.as_dict()
    assert d["reader_workers_per_rank"] == [1] * 8 and d["predicted_bound"] == "host readers"
    assert d["predicted_host_ceiling_windows_per_s"] == 8 * host_plan.READER_RATE["contiguous"]
    assert any("GRANTED" in n for n in d["notes"])
    # ... where the allowed CPUs all sit on socket 0: ranks 4-7 have no local CPUs to be pinned to -> nobody is pinned into a share it cannot run in
    assert all(r.cpus is None for r in small.ranks[4:])
    # the smallest grant at which eight ranks are device-bound on contiguous images: readers x 65 k >= 81 k -> 2 readers + 2 own threads per rank
    need = next(u for u in range(8, 193) if plan_host(list(range(8)), 8, 4096, usable=u, allowed=list(range(192)), shm_free=1 << 40,
                                                      local_cpus=local, storage=storage).as_dict()["predicted_bound"] == "device")
    assert need == 8 * (2 + host_plan.RANK_THREADS)
    # deflated images need more: 81 k / 5.6 k = 15 readers per rank through zlib, 81 k / 12.5 k = 7 through libdeflate
    zneed = next((u for u in range(8, 400) if plan_host(list(range(8)), 16, 4096, usable=u, allowed=list(range(192)), shm_free=1 << 40,
                                                        local_cpus=local, storage=[{"deflate": 16}] * 8).as_dict()["predicted_bound"] == "device"), None)
    import math
    assert zneed == 8 * (math.ceil(81000.0 / host_plan.READER_RATE["deflate"]) + host_plan.RANK_THREADS)
    assert host_plan.READER_RATE["deflate"] in (5600.0, 12500.0)


def test_ram_backed_budget_counts_ram_not_just_tmpfs_space(tmp_path, monkeypatch):
    """A tmpfs reports (nearly) all of RAM as free space; what may be put there is bounded by the RAM the process tree
    may still take -- MemAvailable and the memory cgroup's head-room (both cgroup versions) -- and only half of that.
    (bench.py's multi-rank end-to-end leg, sized by statvfs alone, took a GPU box down in round 5.)"""
    from helen_amd import host_plan
    mi = tmp_path / "meminfo"
    mi.write_text("MemTotal:       131072000 kB\nMemFree:  1 kB\nMemAvailable:   104857600 kB\n")       # 100 GiB
    v2 = tmp_path / "cg2"
    v2.mkdir()
    (v2 / "memory.max").write_text("max\n")
    (v2 / "memory.current").write_text("123\n")
    assert host_plan.ram_available_bytes(str(mi), str(v2)) == 100 << 30          # "max" = no limit
    (v2 / "memory.max").write_text("%d\n" % (64 << 30))
    (v2 / "memory.current").write_text("%d\n" % (4 << 30))
    assert host_plan.ram_available_bytes(str(mi), str(v2)) == 60 << 30           # the cgroup is the tighter bound
    v1 = tmp_path / "cg1"
    (v1 / "memory").mkdir(parents=True)
    (v1 / "memory" / "memory.limit_in_bytes").write_text("9223372036854771712\n")   # version 1's "unlimited"
    (v1 / "memory" / "memory.usage_in_bytes").write_text("1000\n")
    assert host_plan.ram_available_bytes(str(mi), str(v1)) == 100 << 30
    (v1 / "memory" / "memory.limit_in_bytes").write_text("%d\n" % (32 << 30))
    assert host_plan.ram_available_bytes(str(mi), str(v1)) == (32 << 30) - 1000
    assert host_plan.ram_available_bytes(str(tmp_path / "none"), str(tmp_path / "none")) is None
    monkeypatch.setattr(host_plan, "shm_free_bytes", lambda path="/dev/shm": 126 << 30)
    monkeypatch.setattr(host_plan, "ram_available_bytes", lambda: 100 << 30)
    assert host_plan.ram_backed_budget_bytes() == 50 << 30
    monkeypatch.setattr(host_plan, "shm_free_bytes", lambda path="/dev/shm": 10 << 30)
    assert host_plan.ram_backed_budget_bytes() == 10 << 30
    monkeypatch.setattr(host_plan, "ram_available_bytes", lambda: None)
    assert host_plan.ram_backed_budget_bytes() == 10 << 30
    # the bench's two-rank default leg on a 128 GB box whose /dev/shm "has" 126 GB: 300,000 windows per rank (95 GB of
    # RAM-backed files) no longer pass; the leg shrinks to what half of the available RAM takes
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_plan", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, need, where, free = bench.e2e_size(300000, 2, True, free=50 << 30)
    assert where == "/dev/shm" and 8192 <= n < 300000 and need * 1.1 < 50 << 30
    n1, need1, where1, _ = bench.e2e_size(300000, 1, True, free=64 << 30)
    assert n1 == 300000 and where1 == "/dev/shm"
    # round 6: the GPU box's own numbers (300 GiB memory cgroup, 16 CPUs; tmpfs budget = half of it).  The files alone would
    # fit that budget at N = 8 with ~100 k windows per rank; together with the generator workers' transient memory and the
    # ranks' resident memory the leg must stay under 0.8 of the RAM, and N = 1, 2 must not shrink at all
    ram = 300 << 30
    sizes = {w: bench.e2e_size(300000, w, True, free=ram // 2, ram=ram, cpus=16) for w in (1, 2, 4, 8)}
    assert sizes[1][0] == 300000 and sizes[2][0] == 300000
    assert 8192 <= sizes[8][0] < sizes[4][0] < 300000
    for w, (n, need, where, _) in sizes.items():
        workers = w * bench.e2e_generator_processes(w, 16)
        total = need + workers * n / 16.0 * bench.E2E_GENERATOR_TRANSIENT + w * bench.E2E_RANK_RESIDENT_BYTES
        assert where == "/dev/shm" and total <= 0.8 * ram, (w, n, total)
    # RAM unknown (no meminfo, no cgroup): only the tmpfs budget decides
    assert bench.e2e_size(300000, 8, True, free=1 << 40)[0] == 300000


def test_the_plan_prices_the_stitch_stage_of_polish():
    """`polish` runs stitch behind the inference: region decode and overlap alignments want ~2.5 CPUs per rank at the
    device's rate.  Eight ranks under a 16-CPU grant are told so before they start; under 192 CPUs nothing is said."""
    from helen_amd import host_plan
    lookup = lambda d: (None, [])
    small = host_plan.plan_host(list(range(8)), 8, 4096, usable=16, allowed=list(range(16)), shm_free=1 << 40,
                                local_cpus=lookup, stitch=True)
    assert any("HOST-BOUND BY ITS STITCH STAGE" in n for n in small.notes), small.notes
    large = host_plan.plan_host(list(range(8)), 8, 4096, usable=192, allowed=list(range(192)), shm_free=1 << 40,
                                local_cpus=lookup, stitch=True)
    assert not any("STITCH" in n for n in large.notes), large.notes
    plain = host_plan.plan_host(list(range(8)), 8, 4096, usable=16, allowed=list(range(16)), shm_free=1 << 40,
                                local_cpus=lookup)
    assert not any("STITCH" in n for n in plain.notes)


def test_bench_clock_report_reads_hwmon_and_never_raises(tmp_path, monkeypatch):
    """bench.py's `roofline.clock`: the fractions re-read at the sustained clock, and a box without the amdgpu hwmon files
    (this container) reports that instead of taking the bench line down."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_clock", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.with_clock({"bound": "mfma", "frac": 0.48, "path_frac": 0.54}, {"sclk_mhz": 1800.0, "socket_power_w": 1400.0})
    assert r["frac_at_sustained_clock"] == round(0.48 * 2400 / 1800, 4) and r["path_frac_at_sustained_clock"] == round(0.54 * 2400 / 1800, 4)
    r = bench.with_clock({"bound": "mfma", "frac": 0.48}, {"error": "no files"})
    assert "frac_at_sustained_clock" not in r and r["clock"] == {"error": "no files"}
    r = bench.with_clock({"bound": "hbm", "frac": 0.7}, {"sclk_mhz": 1800.0})
    assert "frac_at_sustained_clock" not in r            # an HBM-bound fraction does not scale with the shader clock
    calls = []
    out = bench.sustained_clock(lambda n: calls.append(n), "cuda:0", seconds=0.05)
    assert isinstance(out, dict) and ("error" in out or "sclk_mhz" in out)
