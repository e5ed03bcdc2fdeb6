"""Host-side budget of a multi-GPU `call_consensus` run: how many reader processes each rank gets, which CPUs it is
pinned to, how much RAM-backed slot space all ranks together may take -- and what rate that host can feed.

The reference starts one process per device (models/predict_gpu.py:207-226) and gives each a DataLoader with `-w`
workers, whatever the machine has.  At the device rate of this build (~81 k windows/s per MI355X, fp32) the host is
the part that runs out first: a reader process delivers 27-47 k windows/s when several run (71 k alone;
helen_amd/csrc/h5scan.h, profiles/r03_reader_scaling.txt), so one GPU wants three reader CPUs plus the rank's own two
threads (device stage, writer).  Eight ranks want ~40 CPUs; a container may grant fewer than it shows (cgroup quota:
the 16-CPU box of this project tops out at 240-320 k windows/s of readers, under half of what eight MI355X take), and
a two-socket box has the GPUs split over NUMA nodes.
Nothing here touches torch: the device -> NUMA node map is read from sysfs by PCI address.
"""
import os
import sys

# measured on the GPU box's host (EPYC 9575F under a 16-CPU quota; scripts/reader_bench.py, profiles/r03_reader_scaling.txt):
# one reader process through the direct scanner into a shared slot delivers 71 k windows/s alone, 42-47 k as one of
# four to six, 27-30 k as one of eight to twelve (the quota and memory bandwidth are shared); the planning figure
# is the crowded one.  Device stage of one rank (fp32, 4096-window calls): 81 k.
READER_WINDOWS_PER_S = 30000.0
DEVICE_WINDOWS_PER_S = 81000.0
WRITER_WINDOWS_PER_S = 100000.0
RANK_THREADS = 2            # the rank's own busy threads: device stage + writer (feeder and release threads sleep)
SLOT_BYTES_PER_WINDOW = 90000 + 24000 + 24 + 128 + 2000     # image, positions, meta, contig name, two label rows


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a container
    that sees 256 CPUs may be limited to 16 CPUs' worth of time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())        # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def device_pci_address(device):
    """'dddd:bb:dd.f' of HIP device `device` (torch's device properties), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(int(device))
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def device_local_cpus(pci_address, sysfs="/sys/bus/pci/devices"):
    """(numa node, CPUs local to it) of a PCI device from sysfs; (None, None) when the kernel does not say
    (single-node machines report node -1)."""
    if not pci_address:
        return None, None
    try:
        node = int(open(os.path.join(sysfs, pci_address, "numa_node")).read())
        cpus = parse_cpulist(open(os.path.join(sysfs, pci_address, "local_cpulist")).read())
        if node < 0 or not cpus:
            return None, None
        return node, cpus
    except Exception:
        return None, None


def shm_free_bytes(path="/dev/shm"):
    try:
        if os.path.isdir(path) and os.access(path, os.W_OK):
            st = os.statvfs(path)
            return st.f_bavail * st.f_frsize
    except OSError:
        pass
    return 0


class RankPlan(object):
    """What one rank of predict_gpu gets (picklable: it travels to the spawned process)."""

    def __init__(self, rank, device, reader_workers, cpus, numa_node, slots, slot_prefix):
        self.rank, self.device, self.reader_workers = rank, device, reader_workers
        self.cpus, self.numa_node, self.slots, self.slot_prefix = cpus, numa_node, slots, slot_prefix

    def as_dict(self):
        return {"rank": self.rank, "device": self.device, "reader_workers": self.reader_workers,
                "numa_node": self.numa_node, "cpus_pinned": None if self.cpus is None else len(self.cpus),
                "slots": self.slots}


class HostPlan(object):
    def __init__(self, ranks, usable, requested_workers, shm_free, shm_need, notes):
        self.ranks, self.usable_cpus, self.requested_workers = ranks, usable, requested_workers
        self.shm_free, self.shm_need, self.notes = shm_free, shm_need, notes

    @property
    def host_ceiling(self):
        """windows/s the planned reader processes can deliver (the writer of a rank does more than its device)."""
        return sum(max(1, r.reader_workers) for r in self.ranks) * READER_WINDOWS_PER_S

    @property
    def device_ceiling(self):
        return len(self.ranks) * DEVICE_WINDOWS_PER_S

    def as_dict(self):
        return {"n_ranks": len(self.ranks), "usable_cpus": self.usable_cpus,
                "requested_reader_workers_per_rank": self.requested_workers,
                "reader_workers_per_rank": [r.reader_workers for r in self.ranks],
                "predicted_host_ceiling_windows_per_s": round(self.host_ceiling),
                "predicted_device_ceiling_windows_per_s": round(self.device_ceiling),
                "predicted_bound": "host readers" if self.host_ceiling < self.device_ceiling else "device",
                "shm_free_bytes": self.shm_free, "shm_slot_bytes_all_ranks": self.shm_need,
                "ranks": [r.as_dict() for r in self.ranks], "notes": self.notes}

    def describe(self, out=sys.stderr):
        d = self.as_dict()
        out.write("INFO: HOST PLAN: %d RANK(S), %d USABLE CPUS, READER PROCESSES PER RANK %s (REQUESTED %d), "
                  "HOST CEILING ~%d WINDOWS/S, DEVICE CEILING ~%d WINDOWS/S (%s-BOUND).\n"
                  % (d["n_ranks"], d["usable_cpus"], d["reader_workers_per_rank"], self.requested_workers,
                     d["predicted_host_ceiling_windows_per_s"], d["predicted_device_ceiling_windows_per_s"],
                     d["predicted_bound"].upper()))
        for n in self.notes:
            out.write("INFO: HOST PLAN: " + n + "\n")


def plan_host(devices, num_workers, cap_windows, calls_per_rank=None, usable=None, allowed=None, shm_free=None,
              local_cpus=None, token=None):
    """The plan for `len(devices)` ranks.

    * reader processes per rank = min(requested `-w`, (usable CPUs - RANK_THREADS x ranks) // ranks), at least 1 when
      any were requested: more readers than CPUs only add context switches to a rank whose device stage and writer
      need a CPU each ($HELEN_READERS_UNCAPPED=1 keeps the request);
    * each rank is pinned to the allowed CPUs of its GPU's NUMA node ($HELEN_PIN=0 turns it off; skipped when sysfs has
      no node for the device, when the node's share of the allowed CPUs is smaller than the rank needs, or on a
      single-node machine);
    * slots per rank: five (reader | H2D | kernels | D2H | writer) when the RAM-backed directory has room for all
      ranks' slots with a quarter to spare, else three, else the slots go to the temp directory (SharedSlot falls
      back by itself; the plan only says so beforehand).
    `usable`, `allowed`, `shm_free`, `local_cpus` (device -> (node, cpus)) are injectable for tests."""
    n = len(devices)
    notes = []
    usable = usable_cpus() if usable is None else usable
    if allowed is None:
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    requested = max(0, int(num_workers))
    budget = max(1, (usable - RANK_THREADS * n) // n)
    workers = requested
    if requested > budget and os.environ.get("HELEN_READERS_UNCAPPED", "") != "1":
        workers = budget
        notes.append("%d READER PROCESSES PER RANK REQUESTED, %d GRANTED: %d USABLE CPUS FOR %d RANK(S) OF %d OWN "
                     "THREADS EACH" % (requested, workers, usable, n, RANK_THREADS))
    # NUMA pinning
    pin = os.environ.get("HELEN_PIN", "1") != "0" and n > 1
    cpus_of, node_of = {}, {}
    if pin:
        lookup = local_cpus if local_cpus is not None else (lambda d: device_local_cpus(device_pci_address(d)))
        nodes_seen = set()
        for r, d in enumerate(devices):
            node, cpus = lookup(d)
            node_of[r] = node
            if node is None:
                continue
            nodes_seen.add(node)
            mine = [c for c in cpus if c in set(allowed)]
            cpus_of[r] = mine
        if len(nodes_seen) < 2:
            if nodes_seen or not node_of:
                notes.append("NO NUMA PINNING: ALL DEVICES ON ONE NODE (OR SYSFS HAS NO NODE FOR THEM)")
            cpus_of = {}
        else:
            # a node's allowed CPUs are shared by the ranks whose GPUs sit on it: refuse to pin a rank into a share
            # smaller than what it runs
            per_node = {}
            for r in cpus_of:
                per_node.setdefault(node_of[r], []).append(r)
            for node, rs in per_node.items():
                share = len(cpus_of[rs[0]]) // len(rs)
                if share < workers + RANK_THREADS:
                    notes.append("NODE %d: %d ALLOWED CPUS FOR %d RANK(S) OF %d PROCESSES/THREADS EACH: NOT PINNED"
                                 % (node, len(cpus_of[rs[0]]), len(rs), workers + RANK_THREADS))
                    for r in rs:
                        cpus_of.pop(r)
    # RAM-backed slot budget over all ranks
    shm_free = shm_free_bytes() if shm_free is None else shm_free
    per_slot = int(cap_windows) * SLOT_BYTES_PER_WINDOW
    slots = 5 if calls_per_rank is None else min(5, max(1, int(calls_per_rank)))
    need = n * slots * per_slot
    if need * 1.25 > shm_free and slots > 3:
        slots = 3
        need = n * slots * per_slot
        notes.append("RAM-BACKED SLOTS: THREE PER RANK INSTEAD OF FIVE (%.1f GB FREE IN /dev/shm)" % (shm_free / 1e9))
    if need * 1.05 > shm_free:
        notes.append("RAM-BACKED SLOTS: /dev/shm HAS %.1f GB FREE, %d RANK(S) NEED %.1f GB: SLOTS THAT DO NOT FIT GO TO "
                     "THE TEMP DIRECTORY" % (shm_free / 1e9, n, need / 1e9))
    token = token if token is not None else "%d" % os.getpid()
    ranks = [RankPlan(r, devices[r], workers, cpus_of.get(r), node_of.get(r), slots,
                      "helen_slot_%s_%d_" % (token, r)) for r in range(n)]
    return HostPlan(ranks, usable, requested, shm_free, need, notes)


def apply_rank_plan(plan):
    """In the rank's process, before its readers are spawned (they inherit the mask)."""
    if plan is not None and plan.cpus and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, plan.cpus)
        except OSError as e:
            sys.stderr.write("INFO: RANK %d: CPU PINNING REFUSED (%s).\n" % (plan.rank, e))


def sweep_slots(prefixes, directories=("/dev/shm", None)):
    """Remove slot files a killed rank left behind (its atexit handlers did not run)."""
    import tempfile
    removed = 0
    for d in directories:
        d = d or tempfile.gettempdir()
        try:
            names = os.listdir(d)
        except OSError:
            continue
        for name in names:
            if any(name.startswith(p) for p in prefixes):
                try:
                    os.unlink(os.path.join(d, name))
                    removed += 1
                except OSError:
                    pass
    return removed
