"""Host-side budget of a multi-GPU `call_consensus` run: how many reader processes each rank gets, which CPUs it is
pinned to, how much RAM-backed slot space all ranks together may take -- and what rate that host can feed.

The reference starts one process per device (models/predict_gpu.py:207-226) and gives each a DataLoader with `-w`
workers, whatever the machine has.  At the device rate of this build (~81 k windows/s per MI355X, fp32) the host is
the part that runs out first: a reader process delivers 27-47 k windows/s when several run (71 k alone;
helen_amd/csrc/h5scan.h, profiles/r03_reader_scaling.txt), so one GPU wants three reader CPUs plus the rank's own two
threads (device stage, writer).  Eight ranks want ~40 CPUs; a container may grant fewer than it shows (cgroup quota:
the 16-CPU box of this project tops out at 240-320 k windows/s of readers, under half of what eight MI355X take), and
a two-socket box has the GPUs split over NUMA nodes.
Nothing here touches torch or creates a device context: a device's PCI address is read from the KFD topology in sysfs
(/sys/class/kfd/kfd/topology/nodes/*/properties), its NUMA node from /sys/bus/pci/devices/<address>.
"""
import os
import sys

# measured on the GPU box's host (EPYC 9575F under a 16-CPU quota; scripts/reader_bench.py, profiles/r03_reader_scaling.txt):
# one reader process through the direct scanner into a shared slot delivers 71 k windows/s alone, 42-47 k as one of
# four to six, 27-30 k as one of eight to twelve (the quota and memory bandwidth are shared); the planning figure
# is the crowded one.  Device stage of one rank (fp32, 4096-window calls): 81 k.
READER_WINDOWS_PER_S = 30000.0
# Round 4: what ONE reader (a native thread of helen_io_read_image_runs, or a reader process of the fallback pool) delivers
# depends on how the file stores its images -- profiles/r04_reader_variants.txt (scripts/reader_variants.py on the GPU box's
# host, EPYC 9575F), the figure of one reader among eight: a contiguous image is a copy out of the page cache (92-108 k
# windows/s from one thread, 520-536 k from eight), a chunked one a copy per chunk (65-80 k / 410-450 k), a deflated one is
# zlib's inflate of 90 KB (5.9 k / 45 k on pileup-like pixels; incompressible ones inflate three times faster), and what
# the direct scanner does not take costs libhdf5's six object opens per window (3.2-12.6 k from its one thread per
# process).  The plan prices a run with the storage class it FINDS (the first image of every file:
# helen_amd.native_io.image_storage).
# Round 5: deflated chunks go through libdeflate where the system has it (helen_amd/csrc/h5scan.h; profiles/
# r05_reader_variants.txt: 13.7-16.5 k windows/s from one thread, 100-124 k from eight on pileup-like pixels).
READER_RATE = {"contiguous": 65000.0, "chunked": 50000.0, "deflate": 5600.0, "libhdf5": 5000.0}
try:
    from . import native_io as _native_io
    if _native_io.fast_inflate():
        READER_RATE["deflate"] = 12500.0
except Exception:          # noqa: BLE001 -- no native library: the zlib figure stands
    pass
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


def kfd_gpu_addresses(topology="/sys/class/kfd/kfd/topology/nodes"):
    """PCI addresses ('dddd:bb:dd.f') of the GPUs in the order the ROCm runtime enumerates them: the KFD topology nodes
    with compute units (`simd_count` > 0), by node number; `location_id` is bus << 8 | device << 3 | function, `domain`
    the PCI domain.  No device context is created (the parent of the ranks must not hold the GPUs).  [] when the kernel
    does not expose the topology."""
    out = []
    try:
        nodes = sorted((int(n), n) for n in os.listdir(topology) if n.isdigit())
    except OSError:
        return out
    for _, name in nodes:
        props = {}
        try:
            for line in open(os.path.join(topology, name, "properties")):
                k, _, v = line.strip().partition(" ")
                props[k] = v
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue                     # a CPU node
            loc = int(props["location_id"])
            out.append("%04x:%02x:%02x.%d" % (int(props.get("domain", "0")), (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
        except (KeyError, ValueError):
            out.append(None)
    return out


def visible_gpu_addresses(topology="/sys/class/kfd/kfd/topology/nodes", environ=None):
    """kfd_gpu_addresses filtered the way the runtime filters devices: ROCR_VISIBLE_DEVICES first, then
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES on what is left (lists of indices; anything else -- UUIDs -- gives [])."""
    environ = os.environ if environ is None else environ
    gpus = kfd_gpu_addresses(topology)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if "HIP_VISIBLE_DEVICES" in environ else "CUDA_VISIBLE_DEVICES"):
        text = environ.get(var)
        if text is None or text.strip() == "":
            continue
        try:
            idx = [int(x) for x in text.split(",") if x.strip() != ""]
        except ValueError:
            return []
        if any(i < 0 or i >= len(gpus) for i in idx):
            return []
        gpus = [gpus[i] for i in idx]
    return gpus


def device_pci_address(device, topology="/sys/class/kfd/kfd/topology/nodes"):
    """'dddd:bb:dd.f' of HIP device `device`, from the KFD topology in sysfs (no torch, no device context), or None."""
    gpus = visible_gpu_addresses(topology)
    d = int(device)
    return gpus[d] if 0 <= d < len(gpus) else None


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


def _first_int(path):
    try:
        with open(path) as f:
            t = f.read().split()[0]
        return None if t == "max" else int(t)
    except (OSError, ValueError, IndexError):
        return None


def ram_available_bytes(meminfo="/proc/meminfo", cgroup_root="/sys/fs/cgroup"):
    """RAM this process tree may still take before something is killed: MemAvailable, and the head-room of the
    memory cgroup it runs in (version 2: memory.max - memory.current; version 1: memory.limit_in_bytes -
    memory.usage_in_bytes), whichever is smaller.  tmpfs pages count against both -- a /dev/shm that statvfs reports
    as all of RAM free cannot be filled (a 2-rank bench leg sized by statvfs alone took a GPU box down in round 5).
    None when neither can be read."""
    found = []
    try:
        with open(meminfo) as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    found.append(int(ln.split()[1]) * 1024)
                    break
    except (OSError, ValueError):
        pass
    for limit, usage in (("memory.max", "memory.current"),
                         ("memory/memory.limit_in_bytes", "memory/memory.usage_in_bytes")):
        lim = _first_int(os.path.join(cgroup_root, limit))
        use = _first_int(os.path.join(cgroup_root, usage))
        if lim is not None and use is not None and lim < (1 << 60):
            found.append(max(0, lim - use))
    return min(found) if found else None


def peak_rss_mb(status="/proc/self/status"):
    """High-water mark of this process's resident set (VmHWM) in MB, or None where /proc does not say."""
    try:
        with open(status) as f:
            for ln in f:
                if ln.startswith("VmHWM:"):
                    return int(ln.split()[1]) // 1024
    except (OSError, ValueError, IndexError):
        pass
    return None


def rss_breakdown_mb(status="/proc/self/status"):
    """(anonymous, file-backed + shared-memory) resident MB of this process NOW -- what VmHWM does not tell apart: the pages
    of image files the direct scanner has mapped are resident without being anybody's heap."""
    anon = other = None
    try:
        with open(status) as f:
            for ln in f:
                if ln.startswith("RssAnon:"):
                    anon = int(ln.split()[1]) // 1024
                elif ln.startswith(("RssFile:", "RssShmem:")):
                    other = (other or 0) + int(ln.split()[1]) // 1024
    except (OSError, ValueError, IndexError):
        pass
    return anon, other


def ram_backed_budget_bytes(path="/dev/shm"):
    """What may be PUT into RAM-backed files under `path` by a run: free space of the tmpfs, and no more than half of
    ram_available_bytes() (the other half is for the processes themselves: page-locked slots, page cache, heaps)."""
    free = shm_free_bytes(path)
    ram = ram_available_bytes()
    return free if ram is None else min(free, ram // 2)


class RankPlan(object):
    """What one rank of predict_gpu gets (picklable: it travels to the spawned process)."""

    def __init__(self, rank, device, reader_workers, cpus, numa_node, slots, slot_prefix):
        self.rank, self.device, self.reader_workers = rank, device, reader_workers
        self.cpus, self.numa_node, self.slots, self.slot_prefix = cpus, numa_node, slots, slot_prefix

    def as_dict(self):
        return {"rank": self.rank, "device": self.device, "reader_workers": self.reader_workers,
                "numa_node": self.numa_node, "cpus_pinned": None if self.cpus is None else len(self.cpus),
                "slots": self.slots}


def reader_rate(storage):
    """windows/s of ONE reader over files of the given storage classes: {class: windows} (or files, any weight) -> the
    rate of walking all of them, i.e. the weighted harmonic mean of READER_RATE; None / empty = contiguous."""
    if not storage:
        return READER_RATE["contiguous"]
    total = float(sum(storage.values()))
    if total <= 0:
        return READER_RATE["contiguous"]
    return total / sum(w / READER_RATE.get(c, READER_RATE["libhdf5"]) for c, w in storage.items() if w > 0)


def storage_of_files(files):
    """{storage class: number of files} of a rank's image files (their first images; a file without images counts
    nowhere).  {} when the native reader is not built."""
    from . import native_io
    out = {}
    if not native_io.available():
        return out
    for path in files:
        try:
            c = native_io.image_storage(path)
        except (IOError, OSError):
            c = "libhdf5"
        if c is not None:
            out[c] = out.get(c, 0) + 1
    return out


class HostPlan(object):
    def __init__(self, ranks, usable, requested_workers, shm_free, shm_need, notes, storage=None):
        self.ranks, self.usable_cpus, self.requested_workers = ranks, usable, requested_workers
        self.shm_free, self.shm_need, self.notes = shm_free, shm_need, notes
        self.storage = storage                # per rank: {storage class: files}, or None (not inspected)

    def rank_reader_rate(self, r):
        return reader_rate(self.storage[r] if self.storage else None)

    @property
    def host_ceiling(self):
        """windows/s the planned readers can deliver over the storage they will find (the writer of a rank does more
        than its device)."""
        return sum(max(1, rp.reader_workers) * self.rank_reader_rate(r) for r, rp in enumerate(self.ranks))

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
                "image_storage_per_rank": self.storage,
                "reader_windows_per_s_each": [round(self.rank_reader_rate(r)) for r in range(len(self.ranks))],
                "shm_free_bytes": self.shm_free, "shm_slot_bytes_all_ranks": self.shm_need,
                "ranks": [r.as_dict() for r in self.ranks], "notes": self.notes}

    def describe(self, out=sys.stderr):
        d = self.as_dict()
        out.write("INFO: HOST PLAN: %d RANK(S), %d USABLE CPUS, READERS PER RANK %s (REQUESTED %d) AT ~%s WINDOWS/S EACH, "
                  "HOST CEILING ~%d WINDOWS/S, DEVICE CEILING ~%d WINDOWS/S (%s-BOUND).\n"
                  % (d["n_ranks"], d["usable_cpus"], d["reader_workers_per_rank"], self.requested_workers,
                     sorted(set(d["reader_windows_per_s_each"])),
                     d["predicted_host_ceiling_windows_per_s"], d["predicted_device_ceiling_windows_per_s"],
                     d["predicted_bound"].upper()))
        if self.storage:
            seen = {}
            for st in self.storage:
                for c, k in st.items():
                    seen[c] = seen.get(c, 0) + k
            out.write("INFO: HOST PLAN: IMAGE STORAGE FOUND: %s.\n"
                      % ", ".join("%d FILE(S) %s (~%d WINDOWS/S PER READER)" % (k, c.upper(), READER_RATE.get(c, 0))
                                  for c, k in sorted(seen.items())))
            if self.host_ceiling < self.device_ceiling:
                slow = [c for c in seen if READER_RATE.get(c, 0) < READER_RATE["contiguous"]]
                out.write("WARN: HOST PLAN: THIS RUN IS HOST-BOUND AT ~%d%% OF THE DEVICE RATE%s; MORE READERS (-w) OR "
                          "CONTIGUOUS, UNCOMPRESSED IMAGE FILES WOULD LIFT IT (python -m helen_amd check_images -i <dir> "
                          "--strict SAYS WHICH PATH EVERY FILE TAKES).\n"
                          % (100 * self.host_ceiling / self.device_ceiling,
                             (": THE IMAGES ARE STORED " + " / ".join(c.upper() for c in sorted(slow))) if slow else ""))
        for n in self.notes:
            out.write("INFO: HOST PLAN: " + n + "\n")


# `polish`: what the stitch stage behind the inference costs the host per rank at the device's rate (measured on the GPU
# box's host, 81 k windows/s = 27 k regions/s per MI355X; profiles/r06_stitch_stage_cpu.txt): 0.93 of a CPU over all its
# threads -- region decode, and overlap alignments at ~4 us each since the aligner answers the common join without its
# three passes (70 us and 4.96 CPUs with the shortcuts off; round 5 priced the stage at 2.5)
STITCH_CPUS_PER_RANK = 1.0


def plan_host(devices, num_workers, cap_windows, calls_per_rank=None, usable=None, allowed=None, shm_free=None,
              local_cpus=None, token=None, storage=None, stitch=False):
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
    `storage` = per rank {storage class: files} (storage_of_files) prices the readers with what they will find.
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
    if stitch:
        want = n * (min(workers, 2) + RANK_THREADS + STITCH_CPUS_PER_RANK)
        if want > usable:
            notes.append("STITCH RUNS BEHIND THE INFERENCE AND WANTS ABOUT %.1f CPUS PER RANK AT THE DEVICE'S RATE (REGION DECODE, "
                         "OVERLAP ALIGNMENTS): %d RANK(S) WITH THEIR READERS AND WRITERS WANT %.0f, %d ARE USABLE -- THE RUN "
                         "WILL BE HOST-BOUND BY ITS STITCH STAGE, NOT BY ITS READERS"
                         % (STITCH_CPUS_PER_RANK, n, want, usable))
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
        if any(v is None for v in node_of.values()) and local_cpus is None:
            notes.append("NUMA LOOK-UP: NO PCI ADDRESS / NODE IN SYSFS FOR DEVICE(S) %s"
                         % ",".join(str(devices[r]) for r, v in sorted(node_of.items()) if v is None))
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
    shm_free = ram_backed_budget_bytes() if shm_free is None else shm_free
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
    return HostPlan(ranks, usable, requested, shm_free, need, notes, storage=storage)


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
