"""predict / predict_gpu: the per-device inference loop of `helen call_consensus`.

Same signatures and observable behaviour as helen/modules/python/models/predict_gpu.py:38-226:
one process per device, each with its own model replica and file list, writing
`<output_filename>_<rank>.hdf`.  What differs is the mechanism: the 19-chunk sliding window, the
softmax-accumulate and the argmax of a whole batch run as one helen_polish_host call on the
MI355X; several loader batches are coalesced per call (windows are independent; hidden is zeroed
per window, so coalescing cannot change results), images go up as uint8 through pinned
double-buffered copies and labels come back as uint8; reader processes fill shared-memory slots
while the previous slot is on the GPU and the one before is being written by a writer thread.
No process group is created (the reference's gloo group is never used on this path).

The prediction file is the reference's single `<output>_<rank>.hdf` per rank, written by one thread through the
direct HDF5 emitter of libhelen_io.so (helen_amd/csrc/h5emit.h): libhdf5 itself spends ~100 us creating the
group and three small datasets of a window (8-14 k windows/s per process against ~81 k windows/s of device
throughput), the emitter writes the same objects at ~170 k windows/s.  $HELEN_WRITERS=W > 1 opts into the
round-1 pool of W writer processes: writer 0 keeps `<output>_<rank>.hdf`, writer k > 0 writes
`<output>_<rank>_w<k>.hdf`, all chunks of one region go to the same file, a writer without regions leaves no
file, and stitch takes every `*.hdf` of the directory (StitchInterface.py:35-36).
"""
import collections
import multiprocessing as mp
import os
import queue
import sys
import threading
import time


from .data_store import DataStore
from .host_plan import peak_rss_mb, rss_breakdown_mb
from .options import ImageSizeOptions
from .prediction_writer import prediction_file_name, writer_of_region, writer_process  # noqa: F401
from .sequence_dataset import SequenceDataset, SharedSlot, fill_shared

# windows per device call: scratch is ~4 MB per window, 4096 windows fill 256 CUs x 2 workgroups
DEVICE_CALL_WINDOWS = 4096

# wall seconds per pipeline stage of the last predict() in this process (reported by rank 0)
STAGE_SECONDS = {"read_wait": 0.0, "device": 0.0, "write": 0.0}
# what the last predict() of this process did (windows, seconds, stage seconds, reader processes, ...)
LAST_PREDICT = {}
# threads still handing page-locked slots of finished runs back to the runtime (see predict(): release_device)
BACKGROUND_RELEASE = []

# what the last predict_gpu() of this process did: the host plan and every rank's LAST_PREDICT
LAST_RUN = {}


class _SlotRelease(object):
    """A slot goes back to the readers when EVERY stage that reads its labels is done with it: the writer alone, or --
    `polish` -- the writer and the stitch stage, which work on the same slot side by side (both only read it)."""

    def __init__(self, free_slots, stages):
        self.free_slots, self.stages = free_slots, stages
        self.lock = threading.Lock()
        self.left = {}

    def done(self, slot):
        with self.lock:
            left = self.left.get(id(slot), self.stages) - 1
            if left > 0:
                self.left[id(slot)] = left
                return
            self.left.pop(id(slot), None)
        self.free_slots.put(slot)


def _writer_loop(wq, store, release, err):
    """Writer thread: labels of one device call -> prediction HDF5, then the slot is this stage's no longer."""
    try:
        while True:
            item = wq.get()
            if item is None:
                return
            slot, n, bases, rles = item
            t0 = time.time()
            store.write_batch(slot.contigs[:n], slot.meta[:n], slot.positions[:n], bases, rles)
            STAGE_SECONDS["write"] += time.time() - t0
            release.done(slot)
    except Exception as e:  # surfaced by the caller
        err.append(e)
        release.free_slots.put(None)


def _stitch_loop(sq, stream, release, err):
    """Stitch stage of `polish` (helen_amd.stitch_stream): the regions of a device call are decoded from the slot's label
    buffers -- beside the writer, which stores the same labels -- and their overlap alignments handed to worker threads.
    A failure here (a full spill directory, a dead collector's pipe) is the stitch stage's own: `err` takes it, the stage
    stops decoding, the slots keep circulating and the inference finishes its prediction file -- `polish` then stitches
    that file in a second phase (helen_amd.call_consensus.polish_genome)."""
    failed = False
    while True:
        item = sq.get()
        if item is None:
            return
        slot, n, bases, rles = item
        if not failed:
            try:
                t0 = time.time()
                stream.feed(slot.contigs[:n], slot.meta[:n], slot.positions[:n], bases, rles)
                STAGE_SECONDS["stitch"] = STAGE_SECONDS.get("stitch", 0.0) + time.time() - t0
            except Exception as e:      # noqa: BLE001 -- reported; the run goes on without its pipelined stitch
                failed = True
                err.append(e)
                sys.stderr.write("WARNING: THE STITCH STAGE BEHIND THE INFERENCE STOPPED (%s: %s); THE PREDICTION FILE IS "
                                 "UNAFFECTED AND WILL BE STITCHED AFTER THE RUN.\n" % (type(e).__name__, e))
        release.done(slot)


class _DeviceStage(object):
    """The device stage as a three-stream pipeline: H2D of slot k+1 | kernels of slot k | D2H of slot
    k-1, two device image buffers.  The slots' shared mappings are page-locked in place
    (hipHostRegister, through torch's runtime binding) so both copies are plain DMA from / into the
    memory the readers fill and the writers read -- no staging copy, nothing blocks the host until
    `pop()` waits for the oldest slot's labels.  helen_polish_batch (device pointers, asynchronous
    on the current stream) does the work."""

    def __init__(self, engine, slots, cap, device):
        import torch
        self.torch, self.engine = torch, engine
        self.dev = torch.device("cuda", device)
        self.rt = torch.cuda.cudart()
        self.registered = []
        # (page-locking a fresh shared mapping populates it: 0.09 s per 475 MB slot on the GPU box, 0.5 s for five.
        # Locking the later slots from a helper thread while the first calls run was measured: the set-up shrinks
        # by 0.26 s and the loop grows by as much -- the pages have to be brought in either way)
        for sl in slots:
            if not hasattr(sl, "base_address"):      # a PinnedSlot: page-locked memory of the runtime already
                continue
            ptr, nbytes = sl.base_address()
            rc = self.rt.cudaHostRegister(ptr, nbytes, 0)
            if int(rc) != 0:
                self.close()
                raise RuntimeError("hipHostRegister failed (%s)" % (rc,))
            self.registered.append(ptr)
        self.s_in, self.s_out = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        self.img = [torch.empty((cap, L, H), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.busy = [None, None]          # event after which img[i] may be overwritten
        self.k = 0
        self.inflight = collections.deque()

    def submit(self, slot, n):
        torch = self.torch
        i = self.k & 1
        self.k += 1
        compute = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.s_in):
            if self.busy[i] is not None:
                self.s_in.wait_event(self.busy[i])
            src = slot.images_t[:n] if hasattr(slot, "images_t") else torch.from_numpy(slot.images[:n])
            self.img[i][:n].copy_(src, non_blocking=True)
            ev_in = self.s_in.record_event()
        compute.wait_event(ev_in)
        bases, rles = self.engine.polish(self.img[i][:n])
        ev_c = compute.record_event()
        self.busy[i] = ev_c
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(ev_c)
            pinned = hasattr(slot, "bases_t")
            (slot.bases_t[:n] if pinned else torch.from_numpy(slot.bases[:n])).copy_(bases, non_blocking=True)
            (slot.rles_t[:n] if pinned else torch.from_numpy(slot.rles[:n])).copy_(rles, non_blocking=True)
            ev_out = self.s_out.record_event()
        self.inflight.append((slot, n, ev_out, (bases, rles)))     # keep the device labels alive

    def pop(self):
        slot, n, ev, _ = self.inflight.popleft()
        ev.synchronize()
        return slot, n

    def close(self):
        if self.registered:
            self.torch.cuda.synchronize(self.dev)      # nothing may still be copying from / into the slots
        for ptr in self.registered:
            self.rt.cudaHostUnregister(ptr)
        self.registered = []


class _NativeStage(object):
    """The device stage on the library's own slot pipeline (helen_polish_slot_submit / _wait, include/helen_hip.h): the
    upload of slot k+1, the kernels of slot k and the label download of slot k-1 overlap on three streams INSIDE
    libhelen_hip.so, straight from / into the slots' page-locked memory (helen_host_alloc).  Same interface as
    _DeviceStage, no torch anywhere."""

    def __init__(self, engine):
        self.engine = engine
        self.inflight = collections.deque()

    def _settle(self):
        """Wait for every slot the library still has in flight (oldest first: they complete in order).  A failed submit or
        wait resets the library's pipeline: nothing is in flight any more, whatever this side had counted."""
        for entry in self.inflight:
            if entry[2]:
                entry[2] = False
                try:
                    self.engine.wait()
                except Exception:
                    self._forget()
                    raise

    def _forget(self):
        for entry in self.inflight:
            entry[2] = False
        self.engine.in_flight = 0

    def submit(self, slot, n):
        if getattr(slot, "pinned", False):
            if self.engine.in_flight >= 2:           # (predict()'s loop keeps at most one behind the one just queued)
                self._settle()
            try:
                self.engine.submit(slot.images[:n], slot.bases[:n], slot.rles[:n])
            except Exception:
                self._forget()
                raise
            self.inflight.append([slot, n, True])
        else:                                        # a slot that could not be page-locked: the synchronous staged call
            self._settle()
            self.engine.polish_host(slot.images[:n], out=(slot.bases[:n], slot.rles[:n]))
            self.inflight.append([slot, n, False])

    def pop(self):
        slot, n, queued = self.inflight.popleft()
        if queued:
            try:
                self.engine.wait()
            except Exception:
                self._forget()
                raise
        return slot, n

    def close(self):
        """Tear-down: never raises (a run that failed in the device stage is released here, from a thread of its own)."""
        try:
            self._settle()
        except Exception:       # noqa: BLE001 -- the failure has been reported where it happened
            pass
        self.inflight.clear()


def native_path_wanted():
    """`helen polish` / `call_consensus` run WITHOUT torch by default: the checkpoint is read by helen_amd.checkpoint, the
    device stage is the library's slot pipeline.  $HELEN_DEVICE_STAGE=torch keeps round 4's stage (torch streams and
    pinned tensors), =sync the one synchronous call per slot."""
    return os.environ.get("HELEN_DEVICE_STAGE", "native") in ("native", "async")


_STATE_CACHE = {}


def native_model_state(model_path):
    """(state dict of float32 arrays, hidden_size, gru_layers, epochs) read without torch and checked like
    ModelHandler.load_simple_model + TransducerGRU.load_state_dict do -- or None when this checkpoint needs torch.load."""
    from . import checkpoint
    from .weights import param_shapes
    try:
        st = os.stat(model_path)
        key = (os.path.abspath(model_path), st.st_size, st.st_mtime_ns)
    except OSError:
        return None
    if key not in _STATE_CACHE:
        try:
            _STATE_CACHE.clear()
            _STATE_CACHE[key] = checkpoint.load_simple_model_state(model_path)
        except checkpoint.UnsupportedCheckpoint:
            _STATE_CACHE[key] = None
    got = _STATE_CACHE[key]
    if got is None:
        return None
    state, hidden_size, gru_layers, epochs = got
    if gru_layers != 1:
        raise ValueError("this build implements the shipped HELEN architecture: one bidirectional GRU layer per stage "
                         "(Options.py:27)")
    want = dict(param_shapes(ImageSizeOptions.IMAGE_HEIGHT, hidden_size))
    missing = [k for k in want if k not in state]
    unexpected = [k for k in state if k not in want]
    if missing or unexpected:
        raise RuntimeError("Error(s) in loading state_dict for TransducerGRU: missing %s, unexpected %s" % (missing, unexpected))
    for k, shape in want.items():
        if tuple(state[k].shape) != tuple(shape):
            raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(state[k].shape), tuple(shape)))
    return state, hidden_size, gru_layers, epochs


def writer_count(num_workers):
    """Writer processes per rank: ONE by default -- the reference's single `<output>_<rank>.hdf`
    (predict_gpu.py:55), written by a thread of this process through the direct HDF5 emitter
    (helen_amd/csrc/h5emit.h: ~170 k windows/s, above the device's 81 k).  $HELEN_WRITERS=N > 1 opts into the
    round-1 pool of N writer processes and `<output>_<rank>_w<k>.hdf` shards (useful only with
    HELEN_IO_WRITER=libhdf5, whose ~8 k windows/s per process were the reason for the pool)."""
    env = os.environ.get("HELEN_WRITERS")
    if env:
        return max(1, int(env))
    return 1


class _WriterPool(object):
    """W writer processes; a slot is recycled when all of them are done with it."""

    def __init__(self, output_filename, rank, writers, free_slots, err):
        ctx = mp.get_context("spawn")
        self.writers, self.free_slots, self.err = writers, free_slots, err
        self.done_q = ctx.Queue()
        self.task_qs = [ctx.Queue() for _ in range(writers)]
        self.procs = [ctx.Process(target=writer_process, daemon=True,
                                  args=(k, writers, prediction_file_name(output_filename, rank, k),
                                        self.task_qs[k], self.done_q))
                      for k in range(writers)]
        for p in self.procs:
            p.start()
        self.slots, self.pending, self.t_busy = {}, {}, None
        self.collector = threading.Thread(target=self._collect, daemon=True)
        self.collector.start()

    def submit(self, slot, n):
        self.slots[slot.path] = slot
        self.pending[slot.path] = self.writers
        if self.t_busy is None:
            self.t_busy = time.time()
        for q in self.task_qs:
            q.put((slot.path, slot.cap, n))

    def _collect(self):
        closed = 0
        while closed < self.writers:
            path, error = self.done_q.get()
            if error is not None:
                self.err.append(IOError(error))
                self.free_slots.put(None)
                return
            if path is None:
                closed += 1
                continue
            self.pending[path] -= 1
            if self.pending[path] == 0:
                if not any(self.pending.values()) and self.t_busy is not None:
                    STAGE_SECONDS["write"] += time.time() - self.t_busy
                    self.t_busy = None
                self.free_slots.put(self.slots[path])

    def close(self):
        for q in self.task_qs:
            q.put(None)
        self.collector.join()
        for p in self.procs:
            p.join(timeout=60)
        if self.t_busy is not None:
            STAGE_SECONDS["write"] += time.time() - self.t_busy
            self.t_busy = None


def _feeder_loop(calls, free_slots, ready_q, pool, cap, err, num_workers=0, stop=None):
    """Feeder thread: for each device call take a free slot and get its windows read into it -- by
    the worker pool, in about two contiguous tasks per worker so that every worker has something
    to do whatever the loader batch size is, or inline when num_workers == 0."""
    try:
        for batches in calls:
            slot = free_slots.get()
            if slot is None or (stop is not None and stop.is_set()):   # a writer failed / the run is being torn down
                return
            pairs = [p for b in batches for p in b]
            futures = []
            if pool is not None:
                step = max(16, -(-len(pairs) // max(1, 2 * num_workers)))
                for off in range(0, len(pairs), step):
                    futures.append(pool.submit(fill_shared, slot.path, cap, off, pairs[off:off + step]))
            else:
                fill_shared(slot.path, cap, 0, pairs)
            ready_q.put((slot, len(pairs), futures, len(batches)))
    except Exception as e:
        err.append(e)
    finally:
        ready_q.put(None)           # whatever happened, the consumer gets its end-of-stream


def reader_mode(runs, num_workers, writers):
    """"threads" (the default): the readers are native threads of this process (helen_io_read_image_runs: the direct
    scanner walks the files' mappings in parallel, no interpreter involved) filling page-locked slots -- no reader
    processes to start, no shared-memory files to reserve, page-lock and tear down.  "processes" (round 2's pool over
    shared-memory slots) when libhdf5 has to read some of the files and more than one worker was asked for (the library is
    serialised inside a process), when a writer pool needs slots other processes can attach, or on request
    ($HELEN_READERS=processes)."""
    from . import native_io
    if not native_io.available() or writers > 1:      # (a writer pool attaches the slots by path: no override changes that)
        return "processes"
    want = os.environ.get("HELEN_READERS", "")
    if want in ("threads", "processes"):
        return want
    if num_workers > 1 and any(lib for _, _, lib in runs):
        return "processes"
    return "threads"


_NO_SLOT = object()


def _thread_feeder_loop(calls, free_slots, ready_q, make_slot, n_slots, threads, batch_size, err, device_id, reap_q,
                        stop=None):
    """Feeder of the "threads" mode: per device call a free slot (the first `n_slots` are made here, one at a time, as
    the pipeline asks for them: page-locked allocations cost ~0.1 s each and only the first one is waited for), the
    call's (file, first, count) runs read by `threads` native threads, files the reader has left handed to the reaper."""
    from . import native_io
    try:
        if device_id is not None:                 # (None: the host path, or the native path whose slots name their device)
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(device_id)  # the page-locked allocations below belong to this rank's device
        made = 0
        last_path = None
        for runs in calls:
            if stop is not None and stop.is_set():
                return
            slot = _NO_SLOT
            if made < n_slots:
                try:
                    slot = free_slots.get_nowait()
                except queue.Empty:
                    slot = make_slot()
                    made += 1
            if slot is _NO_SLOT:
                slot = free_slots.get()
            if slot is None or (stop is not None and stop.is_set()):   # a writer failed / the run is being torn down
                return
            n = sum(c for _, _, c in runs)
            lib = native_io.read_image_runs(runs, threads, slot.images[:n], slot.positions[:n], slot.meta[:n],
                                            slot.contigs[:n])
            done = _ReadDone(lib)
            ready_q.put((slot, n, [done], -(-n // batch_size)))
            for path, _, _ in runs:
                if last_path is not None and path != last_path:
                    reap_q.put(last_path)
                last_path = path
    except Exception as e:
        err.append(e)
    finally:
        ready_q.put(None)           # whatever happened, the consumer gets its end-of-stream
        reap_q.put(None)


class _ReadDone(object):
    """What a finished read hands the main loop in place of the pool's futures."""

    def __init__(self, through_library):
        self.through_library = through_library

    def result(self):
        return self


def _reaper_loop(reap_q):
    """Lets go of the mappings of files the reader has moved past (unmapping gigabytes of touched pages is tenths of a
    second of kernel time: here it runs beside the device, not at the end of the run)."""
    from . import native_io
    while True:
        path = reap_q.get()
        if path is None:
            return
        native_io.forget_images(path)


def _get_or_error(q, *error_lists):
    """q.get() that gives up (returns None, like the end-of-stream sentinel) as soon as a pipeline stage has
    reported an error: a failed writer must fail the run, not leave it waiting for a slot that never comes."""
    while True:
        try:
            return q.get(timeout=0.5)
        except queue.Empty:
            if any(error_lists):
                return None


def _remove_stale_outputs(output_filename, rank):
    """A reused output directory may hold `<output>_<rank>.hdf` / `<output>_<rank>_w<k>.hdf` of an earlier run with
    another writer count; stitch takes every *.hdf of the directory (StitchInterface.py:35-36) and would merge them
    in.  The reference truncates its one file (DataStore mode 'w'); this rank's shards get the same treatment."""
    import glob
    stale = [prediction_file_name(output_filename, rank)] + \
        glob.glob(glob.escape(output_filename + "_" + str(rank)) + "_w*.hdf")
    for path in stale:
        if os.path.isfile(path):
            os.unlink(path)


# what the stitch stage of the last predict() of this process collected (helen_amd.stitch_stream.StreamResult), or None
LAST_STREAM = [None]


def _discard_early_engine(early_engine):
    """Set-up failed after the early engine thread was started: wait for it and release what it made."""
    if not early_engine:
        return
    t = early_engine.get("thread")
    if t is not None:
        t.join()
    eng = early_engine.pop("engine", None)
    if eng is not None:
        try:
            eng.close()
        except Exception:       # noqa: BLE001 -- already failing: the first error is the one to report
            pass


def predict(test_file, output_filename, model_path, batch_size, num_workers, rank, device_id, plan=None, cpu_threads=None,
            stitch_threads=None, stitch_export=None):
    """Run inference over the image files `test_file` (a list) on device `device_id` and write
    `<output_filename>_<rank>.hdf` (predict_gpu.py:38-179).  `plan` (helen_amd.host_plan.RankPlan, from
    predict_gpu) caps the reader processes at what the host grants this rank and names its slots.
    `cpu_threads` (an int) makes it the predict of a run WITHOUT --gpu_mode (models/predict_cpu.py:39-170): the same
    pipeline around the host engine (helen_amd.cpu_engine, `cpu_threads` OpenMP threads), one loader batch per call.

    Pipeline over shared-memory slots of one device call each:
      reader processes fill slot k+2 | H2D k+1 | kernels k | D2H k-1 | writer(s) store slot k-2."""
    from . import native_io
    start_time = time.time()
    # Without torch by default (helen_amd.native_engine): decided HERE, before anything of the device is touched -- a process
    # on the native path runs on the system's HIP runtime and must not import torch's afterwards.
    native = native_model_state(model_path) if native_path_wanted() else None
    t_model_file = time.time() - start_time
    if native is None:
        import torch
    if plan is not None:
        num_workers = min(num_workers, plan.reader_workers) if num_workers > 0 else 0
    # The device context and the model (0.2-0.3 s: the runtime's start-up, 15.9 GB of scratch) are created by a thread of
    # their own from HERE, beside the indexing of the image files and the start of the readers, instead of after them:
    # every entry of the C ABI selects the model's device itself, so the thread that creates the model need not be the
    # one that uses it.
    early_engine = None
    if native is not None and cpu_threads is None:
        early_engine = {"t0": time.time()}

        def create_engine_early():
            try:
                from .native_engine import NativeEngine
                cap_ = max(1, DEVICE_CALL_WINDOWS // batch_size) * batch_size
                early_engine["engine"] = NativeEngine(native[0], device=device_id, max_windows=min(DEVICE_CALL_WINDOWS, cap_),
                                                      precision=os.environ.get("HELEN_PRECISION", "fp32"))
            except BaseException as e:          # noqa: BLE001 -- raised where the engine is needed
                early_engine["error"] = e
            early_engine["took"] = time.time() - early_engine["t0"]
        early_engine["thread"] = threading.Thread(target=create_engine_early, daemon=True)
        early_engine["thread"].start()
    # (everything up to the try block of the loop below is set-up that can fail -- a bad image file, an unwritable output:
    # the early engine's thread and its device model must not outlive such a failure in a long-lived process)
    try:
        native_io.close_readers()          # a long-lived process may have these paths mapped from an earlier run
        _remove_stale_outputs(output_filename, rank)
        writers = writer_count(num_workers)
        prediction_data_file = DataStore(prediction_file_name(output_filename, rank), mode="w") \
            if writers == 1 else None
        on_host = cpu_threads is not None
        group = 1 if on_host else max(1, DEVICE_CALL_WINDOWS // batch_size)       # loader batches per engine call
        cap = group * batch_size

        # Readers and writers first: their processes start (and the first slots fill) while the model is
        # loaded and the device context is created below.
        test_data = SequenceDataset(image_directory=None, file_list=test_file)
        total_windows = len(test_data)
        total_batches = -(-total_windows // batch_size)
        mode = reader_mode(test_data.runs, num_workers, writers)
        free_slots, ready_q, wq = queue.Queue(), queue.Queue(maxsize=2), queue.Queue()
        pool = None
        if mode == "threads":
            calls = test_data.call_runs(cap)
            # reader filling | H2D | kernels | D2H | writer: five slots keep every stage busy
            n_slots = min(plan.slots if plan is not None else 5, max(1, len(calls)))
            slots = []

            def make_slot():
                from .sequence_dataset import NativeSlot, PinnedSlot
                slots.append(NativeSlot(cap, device_id, pin=not on_host) if native is not None else PinnedSlot(cap, pin=not on_host))
                return slots[-1]
        else:
            pairs = test_data.all_images
            batches = [pairs[i:i + batch_size] for i in range(0, len(pairs), batch_size)]   # sequential,
            calls = [batches[i:i + group] for i in range(0, len(batches), group)]           # short last batch
            n_slots = min(plan.slots if plan is not None else 5, max(1, len(calls)))
            slots = [SharedSlot(cap, prefix=plan.slot_prefix if plan is not None else "helen_slot_") for _ in range(n_slots)]
            for sl in slots:
                free_slots.put(sl)
            if num_workers > 0 and calls:
                import concurrent.futures as cf
                pool = cf.ProcessPoolExecutor(num_workers, mp_context=mp.get_context("spawn"))
        ferr, werr, serr = [], [], []          # failures of the feeder, of the writer, and of the stitch stage (its own: not fatal)
        reaper = None
        stop_feeding = threading.Event()
        # `polish`: stitch runs behind the inference (stitch_threads = its -t); only with the one-file writer of this process
        stream = sq = stitcher = None
        LAST_STREAM[0] = None
        if stitch_threads is not None and writers == 1:
            from . import stitch_stream
            if stitch_stream.enabled():
                export = None
                if stitch_export is not None:       # a multi-rank run: the regions go to the collectors (helen_amd.stitch_collect)
                    from .stitch_collect import RegionExport
                    export = RegionExport(stitch_export[0], rank, stitch_export[1])
                stream = stitch_stream.RegionStream(prediction_file_name(output_filename, rank), stitch_threads, export=export)
                sq = queue.Queue()
        if mode == "threads":
            reap_q = queue.Queue()
            reaper = threading.Thread(target=_reaper_loop, args=(reap_q,), daemon=True)
            reaper.start()
            feeder = threading.Thread(target=_thread_feeder_loop, daemon=True,
                                      args=(calls, free_slots, ready_q, make_slot, n_slots, max(1, num_workers), batch_size,
                                            ferr, None if on_host or native is not None else device_id, reap_q, stop_feeding))
        else:
            feeder = threading.Thread(target=_feeder_loop, daemon=True,
                                      args=(calls, free_slots, ready_q, pool, cap, ferr, num_workers, stop_feeding))
        for k in STAGE_SECONDS:
            STAGE_SECONDS[k] = 0.0
        if writers == 1:
            release = _SlotRelease(free_slots, 2 if stream is not None else 1)
            writer = threading.Thread(target=_writer_loop, args=(wq, prediction_data_file, release, werr), daemon=True)
            writer.start()
            writer_pool = None
            if stream is not None:
                stitcher = threading.Thread(target=_stitch_loop, args=(sq, stream, release, serr), daemon=True)
                stitcher.start()
        else:
            writer_pool = _WriterPool(output_filename, rank, writers, free_slots, werr)
        feeder.start()

        stage = engine = None

        def to_writer(slot, n):
            if writer_pool is None:
                item = (slot, n, slot.bases[:n], slot.rles[:n])
                wq.put(item)
                if sq is not None:
                    sq.put(item)
            else:
                writer_pool.submit(slot, n)

    except BaseException:
        _discard_early_engine(early_engine)
        raise
    through_library = 0
    t_setup = t_loop_end = start_time
    batch_iterator = 0
    setup_took = {}
    try:
        t_s = time.time()
        if native is not None:
            setup_took["MODEL FILE"] = t_model_file        # (read first of all, before the readers started)
            if on_host:
                from .cpu_engine import CpuEngine
                engine = CpuEngine(native[0], threads=cpu_threads)
            else:
                early_engine["thread"].join()
                if "error" in early_engine:
                    raise early_engine["error"]
                engine = early_engine.pop("engine")
                setup_took["DEVICE CONTEXT + WEIGHTS + ENGINE (BESIDE THE INDEXING) %.2f, WAITED" % early_engine["took"]] = \
                    time.time() - t_s
            if on_host:
                setup_took["ENGINE"] = time.time() - t_s
        else:
            from .model_handler import ModelHandler
            transducer_model, hidden_size, gru_layers, prev_ite = ModelHandler.load_simple_model(
                model_path, input_channels=ImageSizeOptions.IMAGE_CHANNELS,
                image_features=ImageSizeOptions.IMAGE_HEIGHT, seq_len=ImageSizeOptions.SEQ_LENGTH,
                num_base_classes=ImageSizeOptions.TOTAL_BASE_LABELS,
                num_rle_classes=ImageSizeOptions.TOTAL_RLE_LABELS)
            transducer_model.eval()
            setup_took["MODEL FILE"] = time.time() - t_s
            t_s = time.time()
            if on_host:
                transducer_model.use_cpu(cpu_threads)
            else:
                torch.cuda.set_device(device_id)
                transducer_model.to(device_id)
            setup_took["DEVICE CONTEXT + WEIGHTS"] = time.time() - t_s
            t_s = time.time()
            transducer_model.set_capacity(min(DEVICE_CALL_WINDOWS, cap))
            engine = transducer_model.engine
            setup_took["ENGINE"] = time.time() - t_s
        if rank == 0:
            print(prediction_file_name(output_filename, rank))
            if on_host:
                sys.stderr.write("INFO: HOST PATH (NO --gpu_mode), %d THREAD(S) PER CALLER.\n" % max(1, cpu_threads))
            else:
                sys.stderr.write("INFO: MI355X HIP PATH, DEVICE " + str(device_id) + ", "
                                 + str(engine.device_bytes >> 20) + " MiB OF DEVICE MEMORY HELD.\n")
            sys.stderr.write("Loading data\n")
        if native is not None and not on_host and calls:
            stage = _NativeStage(engine)
        elif native is None and os.environ.get("HELEN_DEVICE_STAGE", "async") != "sync" and hasattr(engine, "polish") \
                and not on_host and torch.cuda.is_available() and calls:
            t_s = time.time()
            try:
                stage = _DeviceStage(engine, slots, cap, device_id)
            except Exception as e:      # page-locking refused (ulimit -l, container policy): staged copies
                sys.stderr.write("INFO: SLOTS NOT PAGE-LOCKED (" + str(e) + "), USING STAGED COPIES.\n")
            setup_took["PAGE-LOCKS + STREAMS"] = time.time() - t_s
        t_setup = time.time()
        while True:
            t0 = time.time()
            item = _get_or_error(ready_q, werr, ferr)
            if item is None:
                break
            slot, n, futures, nb = item
            for f in futures:                    # raises the reader's exception, if any
                through_library += getattr(f.result(), "through_library", 0)
            t1 = time.time()
            if stage is not None:
                stage.submit(slot, n)
                while len(stage.inflight) > 1:   # keep one call in flight behind the one just queued
                    to_writer(*stage.pop())
            else:
                engine.polish_host(slot.images[:n], out=(slot.bases[:n], slot.rles[:n]))
                to_writer(slot, n)
            STAGE_SECONDS["read_wait"] += t1 - t0
            STAGE_SECONDS["device"] += time.time() - t1
            batch_iterator += nb
            if rank == 0:
                eta = (time.time() - start_time) / batch_iterator * (total_batches - batch_iterator)
                sys.stderr.write("INFO: BATCHES DONE: %d/%d. ESTIMATED TIME LEFT: %d MINS %d SECS.\n"
                                 % (batch_iterator, total_batches, int(eta // 60), int(eta) % 60))
            if werr:
                break
        t1 = time.time()
        while stage is not None and stage.inflight and not werr:
            to_writer(*stage.pop())
        STAGE_SECONDS["device"] += time.time() - t1
        t_loop_end = time.time()
    finally:
        # Tear-down in parallel with the writer's last slots and the file's close (the group structures of the
        # prediction file are written there): the reader pool, the page-locks, the 16 GB of device scratch and the
        # shared-memory slots each take 0.1-0.3 s to let go of.
        took = {}
        writer_done = threading.Event()
        if werr or ferr or sys.exc_info()[0] is not None:
            # the loop ended early: a feeder blocked on the bounded ready queue or waiting for a slot would never see
            # its end -- take what it has queued, hand it the give-up sentinel, and let it (and the reaper it feeds) finish
            stop_feeding.set()
            free_slots.put(None)
            deadline = time.time() + 30.0
            while feeder.is_alive() and time.time() < deadline:
                try:
                    ready_q.get(timeout=0.05)
                except queue.Empty:
                    pass
            feeder.join(timeout=1.0)
            if reaper is not None:
                reaper.join(timeout=10.0)

        def release_device():
            t = time.time()
            if stage is not None:
                stage.close()
            took["PAGE-LOCKS"] = time.time() - t
            t = time.time()
            if engine is not None:
                engine.close()
            took["DEVICE MEMORY"] = time.time() - t
            writer_done.wait()          # the writer reads labels and positions out of the slots
            t = time.time()
            # Page-locked slots of this process (1.9 GB; the runtime takes 0.05 s to let go of each) are released
            # BEHIND the caller's back: nobody waits for them -- `polish` goes on to the joins and the FASTA, a command
            # ends (helen_amd.cli.leave) -- and a caller that must know joins BACKGROUND_RELEASE.  Slots that are files
            # (reader processes) are removed here and now.
            later = [sl for sl in slots if not hasattr(sl, "path") or sl.path is None]
            for sl in slots:
                if sl not in later:
                    sl.close()
            if later:
                def free_slots_later(them=later):
                    for sl in them:
                        sl.close()
                th = threading.Thread(target=free_slots_later, daemon=True)
                BACKGROUND_RELEASE[:] = [x for x in BACKGROUND_RELEASE if x.is_alive()] + [th]
                th.start()
            took["SLOTS"] = time.time() - t

        def release_readers():
            t = time.time()
            pool.shutdown(wait=True, cancel_futures=True)
            took["READERS"] = time.time() - t
        side = [threading.Thread(target=release_device, daemon=True)]
        if pool is not None:
            side.append(threading.Thread(target=release_readers, daemon=True))
        if writer_pool is None:
            wq.put(None)
            if sq is not None:
                sq.put(None)
        for t in side:
            t.start()
        t_side = time.time()
        if writer_pool is None:
            writer.join()
            if stitcher is not None:
                stitcher.join()
        else:
            writer_pool.close()
        t_drained = time.time()
        writer_done.set()
        close_error = None
        if prediction_data_file is not None and not (ferr or werr):
            try:
                prediction_data_file.close()
            except Exception as e:      # surfaced below, after the tear-down
                close_error = e
        t_closed = time.time()
        for t in side:
            t.join()
        if stream is not None and (ferr or werr or close_error is not None or sys.exc_info()[0] is not None):
            stream.abort()
    if ferr:
        raise ferr[0]
    if werr:
        raise werr[0]
    if close_error is not None:
        raise close_error
    stitch_wait = 0.0
    stitch_failed = None
    if stream is not None and serr:
        stream.abort()
        stitch_failed = "%s: %s" % (type(serr[0]).__name__, serr[0])
        stream = None
    if stream is not None:
        t_s = time.time()
        try:
            LAST_STREAM[0] = stream.finish()        # the overlap alignments still queued with the worker threads
        except Exception as e:      # noqa: BLE001 -- the stitch stage's own failure: the prediction file is complete
            stream.abort()
            stitch_failed = "%s: %s" % (type(e).__name__, e)
            sys.stderr.write("WARNING: THE STITCH STAGE BEHIND THE INFERENCE FAILED AT ITS END (%s); THE PREDICTION FILE IS "
                             "UNAFFECTED AND WILL BE STITCHED AFTER THE RUN.\n" % stitch_failed)
            stream = None
        stitch_wait = time.time() - t_s
    LAST_PREDICT.clear()
    LAST_PREDICT.update({
        "rank": rank, "device": device_id, "windows": total_windows, "seconds": round(time.time() - start_time, 3),
        "stage_seconds": {k: round(v, 3) for k, v in STAGE_SECONDS.items()},
        "setup_seconds": round(t_setup - start_time, 3), "close_seconds": round(time.time() - t_loop_end, 3),
        "reader_workers": num_workers, "reader_mode": mode, "slots": n_slots, "device_calls": len(calls),
        "stitch_stream": None if stream is None else dict(LAST_STREAM[0].stats, wait_seconds=round(stitch_wait, 3)),
        "stitch_failed": stitch_failed, "peak_rss_mb": peak_rss_mb(), "rss_anon_mb": rss_breakdown_mb()[0],
        "cpus_pinned": None if plan is None or not plan.cpus else len(plan.cpus),
        "numa_node": None if plan is None else plan.numa_node})
    if rank == 0:
        sys.stderr.write("INFO: %d WINDOWS IN %.1f SECS (WAITING FOR READERS %.1f, DEVICE %.1f, WRITER BUSY %.1f%s; "
                         "MODEL + ENGINE SET-UP %.1f [%s], FLUSH + CLOSE %.1f = LAST SLOTS %.2f + FILE CLOSE %.2f + "
                         "RELEASE %.2f [%s]).\n"
                         % (total_windows, time.time() - start_time, STAGE_SECONDS["read_wait"],
                            STAGE_SECONDS["device"], STAGE_SECONDS["write"],
                            "" if stream is None else ", STITCH STAGE BUSY %.1f" % STAGE_SECONDS.get("stitch", 0.0),
                            t_setup - start_time,
                            ", ".join("%s %.2f" % kv for kv in setup_took.items()),
                            time.time() - t_loop_end, t_drained - t_side, t_closed - t_drained,
                            time.time() - t_closed,
                            ", ".join("%s %.2f" % kv for kv in sorted(took.items()))))
        if through_library:
            sys.stderr.write("INFO: %d OF THEM WERE READ THROUGH LIBHDF5: THE DIRECT IMAGE SCANNER DOES NOT TAKE THEIR "
                             "STORAGE (CHUNKED / FILTERED / NEW-STYLE FILE); SEE python -m helen_amd check_images.\n"
                             % through_library)


def _setup_cpu(rank, total_callers, args, all_input_files, result_q=None):
    """One caller of a run without --gpu_mode (predict_cpu.py:177-195; no process group is created: it was never used)."""
    output_filepath, model_path, batch_size, num_workers, threads, stitch_threads = args[:6]
    stitch_export = args[6] if len(args) > 6 else None
    if result_q is not None:
        import signal

        def on_term(signum, frame):
            raise SystemExit(143)
        signal.signal(signal.SIGTERM, on_term)
        try:
            os.setpgid(0, 0)
        except OSError:
            pass
    predict(all_input_files[rank], output_filepath, model_path, batch_size, num_workers, rank, None, cpu_threads=threads,
            stitch_threads=stitch_threads, stitch_export=stitch_export)
    if result_q is not None:
        result_q.put((rank, _rank_report()))


def _rank_report():
    """What a spawned rank sends its parent: LAST_PREDICT, with the stitch stage's regions parked in a file."""
    info = dict(LAST_PREDICT)
    if LAST_STREAM[0] is not None:
        if not LAST_STREAM[0].stats.get("exported"):      # (exported: the collectors have the regions, helen_amd.stitch_collect)
            from .stitch_stream import spill_directory
            size = sum(len(v) for v in LAST_STREAM[0].regions.values() if v is not None)
            info["stream_file"] = LAST_STREAM[0].save(spill_directory(size) or os.path.dirname(os.path.abspath(LAST_STREAM[0].file)))
        LAST_STREAM[0] = None
    return info


def _collect_streams(in_process, rank_infos):
    """The StreamResults of a run (helen_amd.stitch_stream): this process's own, or the files its ranks left."""
    from .stitch_stream import StreamResult
    if in_process:
        return [LAST_STREAM[0]] if LAST_STREAM[0] is not None else None
    paths = [r.pop("stream_file", None) for r in rank_infos]
    if not paths or any(p is None for p in paths):
        for p in paths:
            if p is not None:
                try:
                    os.unlink(p)
                except OSError:
                    pass
        return None
    return [StreamResult.load(p) for p in paths]


def _start_collectors(output_filepath, total_callers, stitch_threads, num_workers, file_chunks=None):
    """`polish` over several ranks: the collector processes that take the ranks' regions (helen_amd.stitch_collect), or
    None (no stitch behind this run: `call_consensus` alone, $HELEN_STITCH_PIPELINE=0, or a pool of writer processes --
    a rank streams its regions only beside the one-file writer, and a collector waits for every rank's end marker)."""
    if stitch_threads is None or writer_count(num_workers) != 1:
        return None
    from . import stitch_stream
    if not stitch_stream.enabled():
        return None
    from .stitch_collect import CollectorRun
    # two bytes per window position parked between ranks and collectors; a window is at most 114 KB of an image file
    # (fewer when the file is deflated: the floor of spill_directory and the run-time fallback cover that)
    expected = 0
    for chunk in file_chunks or []:
        for path in chunk:
            try:
                expected += os.path.getsize(path) // 114000 * 2 * ImageSizeOptions.SEQ_LENGTH
            except OSError:
                pass
    return CollectorRun([prediction_file_name(output_filepath, r) for r in range(total_callers)], stitch_threads,
                        expected_bytes=expected).start()


def _streams_of_run(collectors, results, failed, total_callers, what):
    """What a multi-rank run hands `polish`: the collectors (they hold the regions), the ranks' own streams (a run without
    collectors), or None; raises when a rank failed."""
    if failed:
        if collectors is not None:
            collectors.abort()
        for r in results.values():
            p = r.pop("stream_file", None)
            if p is not None:
                try:
                    os.unlink(p)
                except OSError:
                    pass
        raise RuntimeError("prediction process(es) failed: " + ", ".join(
            "rank %d exit %s" % f for f in failed) + "; the other %s were terminated" % what)
    stopped = sorted(r for r, info in results.items() if info.get("stitch_failed"))
    if stopped:         # a rank's stitch stage stopped (its prediction file is complete): no pipelined result for this run
        sys.stderr.write("WARNING: THE STITCH STAGE OF RANK(S) %s STOPPED (%s): STITCHING THE PREDICTION FILES AFTER THE RUN INSTEAD.\n"
                         % (", ".join(str(r) for r in stopped), results[stopped[0]]["stitch_failed"]))
        if collectors is not None:
            collectors.abort()
        for r in results.values():
            p = r.pop("stream_file", None)
            if p is not None:
                try:
                    os.unlink(p)
                except OSError:
                    pass
        return None
    if collectors is not None:
        for r in results.values():                     # (the ranks' stubs: statistics only)
            p = r.pop("stream_file", None)
            if p is not None:
                try:
                    os.unlink(p)
                except OSError:
                    pass
        if len(results) != total_callers:
            collectors.abort()
            return None
        return collectors
    return _collect_streams(False, [results[r] for r in sorted(results)]) if len(results) == total_callers else None


def predict_cpu(file_chunks, output_filepath, model_path, batch_size, total_callers, threads, num_workers,
                stitch_threads=None):
    """`callers` processes, each over its own file list with `threads` threads, each writing `<output>_<rank>.hdf`
    (models/predict_cpu.py:198-248).  The engine is libhelen_cpu.so (no ONNX export: the .pkl is all it needs); a failing
    caller takes the others down, as mp.spawn(join=True) does (:245-248)."""
    args = (output_filepath, model_path, batch_size, num_workers, max(1, int(threads)), stitch_threads, None)
    LAST_RUN.clear()
    LAST_RUN.update({"host_plan": None, "ranks": []})
    t0 = time.time()
    if total_callers == 1:
        _setup_cpu(0, 1, args, file_chunks)
        LAST_RUN["ranks"] = [dict(LAST_PREDICT)]
        LAST_RUN["seconds"] = round(time.time() - t0, 3)
        return _collect_streams(True, None)
    collectors = _start_collectors(output_filepath, total_callers, stitch_threads, num_workers, file_chunks)
    if collectors is not None:
        args = args[:5] + (max(1, int(stitch_threads) // total_callers), collectors.export_spec())
    try:
        results, failed = run_ranks(_setup_cpu, [(r, total_callers, args, file_chunks) for r in range(total_callers)])
    except BaseException:
        if collectors is not None:
            collectors.abort()
        raise
    LAST_RUN["ranks"] = [results[r] for r in sorted(results)]
    LAST_RUN["seconds"] = round(time.time() - t0, 3)
    return _streams_of_run(collectors, results, failed, total_callers, "callers")


def _setup(rank, total_callers, args, all_input_files, all_devices, plans=None, result_q=None):
    output_filepath, model_path, batch_size, num_workers, stitch_threads = args[:5]
    stitch_export = args[5] if len(args) > 5 else None
    plan = plans[rank] if plans is not None else None
    if result_q is not None:
        # a spawned rank: SIGTERM from the parent (a sibling failed) must unwind through predict()'s tear-down --
        # reader pool, page-locks, slot files -- instead of leaving them behind
        import signal

        def on_term(signum, frame):
            raise SystemExit(143)
        signal.signal(signal.SIGTERM, on_term)
        try:
            os.setpgid(0, 0)          # own process group: the parent's last resort is killpg
        except OSError:
            pass
    from .host_plan import apply_rank_plan
    apply_rank_plan(plan)
    predict(all_input_files[rank], output_filepath, model_path, batch_size, num_workers, rank,
            all_devices[rank], plan=plan, stitch_threads=stitch_threads, stitch_export=stitch_export)
    if result_q is not None:
        result_q.put((rank, _rank_report()))


def run_ranks(target, argsets, slot_prefixes=(), grace_seconds=10.0):
    """Start one spawned process per entry of `argsets` running `target(*args, result_q)`, wait for ALL of them at
    once, and when one exits non-zero terminate the others: SIGTERM first (a rank turns it into its normal
    tear-down), SIGKILL to the rank's process group after `grace_seconds`, then sweep the slot files a killed rank
    could not remove.  -> ({rank: what it put on result_q}, [(rank, exit code), ...])."""
    from multiprocessing.connection import wait

    from .host_plan import sweep_slots
    ctx = mp.get_context("spawn")
    result_q = ctx.Queue()
    procs = [ctx.Process(target=target, args=tuple(a) + (result_q,)) for a in argsets]
    for p in procs:
        p.start()
    failed, alive = [], {p.sentinel: (r, p) for r, p in enumerate(procs)}
    results = {}

    def drain():
        while True:
            try:
                r, info = result_q.get_nowait()
            except queue.Empty:
                return
            results[r] = info

    def take_down(why):
        """SIGTERM, the grace period, SIGKILL to the process group, the slot sweep -- for every rank still alive."""
        if not alive:
            return
        sys.stderr.write("ERROR: %s: TERMINATING THE OTHER %d RANK(S).\n" % (why, len(alive)))
        for r, p in alive.values():
            p.terminate()
        deadline = time.time() + grace_seconds
        for r, p in alive.values():
            p.join(max(0.1, deadline - time.time()))
            if p.is_alive():
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    p.kill()
                p.join()
        alive.clear()
        sweep_slots(list(slot_prefixes))

    # The ranks sit in process groups of their own (_setup), so a Ctrl-C at the terminal or a scheduler's SIGTERM reaches
    # only this process: it must pass the signal on, or the ranks would keep the GPUs while this process hangs in
    # multiprocessing's exit handler (the reference's mp.spawn children share the parent's group and die with it,
    # predict_gpu.py:223).  SIGTERM becomes SystemExit here for the duration of the wait.
    import signal

    def on_term(signum, frame):
        raise SystemExit(128 + signum)
    previous = None
    if threading.current_thread() is threading.main_thread():
        previous = signal.signal(signal.SIGTERM, on_term)
    try:
        while alive:
            for sentinel in wait(list(alive), timeout=0.5):
                r, p = alive.pop(sentinel)
                p.join()
                if p.exitcode != 0:
                    failed.append((r, p.exitcode))
            drain()
            if failed and alive:
                take_down("RANK %d EXITED WITH %s" % failed[0])
    except BaseException as e:          # KeyboardInterrupt, SystemExit (SIGTERM), anything unexpected in the wait
        take_down("INTERRUPTED (%s)" % type(e).__name__)
        raise
    finally:
        if previous is not None:
            signal.signal(signal.SIGTERM, previous)
    time.sleep(0.05)
    drain()
    return results, failed


def predict_gpu(file_chunks, output_filepath, model_path, batch_size, total_callers, devices,
                num_workers, stitch_threads=None):
    """One process per device, each over its own file list (predict_gpu.py:207-226).  A failing child raises
    here AND takes its siblings down, as mp.spawn(join=True) does (predict_gpu.py:223): the parent waits on all
    ranks at once, and the first non-zero exit terminates the others (SIGTERM, which a rank turns into its normal
    tear-down; SIGKILL to its process group after ten seconds; slot files swept).

    Before anything starts the host is budgeted over all ranks (helen_amd.host_plan): reader processes per rank
    from the usable CPUs, NUMA pinning of each rank to its GPU's node, one RAM-backed slot budget."""
    from .host_plan import plan_host, storage_of_files
    total_stitch_threads = stitch_threads
    if stitch_threads is not None:          # `polish`: each rank gets its share of the stitch threads
        stitch_threads = max(1, int(stitch_threads) // max(1, total_callers))
    args = (output_filepath, model_path, batch_size, num_workers, stitch_threads)
    group = max(1, DEVICE_CALL_WINDOWS // batch_size)
    host = plan_host(list(devices[:total_callers]), num_workers, group * batch_size,
                     storage=[storage_of_files(file_chunks[r]) for r in range(total_callers)],
                     stitch=total_stitch_threads is not None)
    host.describe()
    LAST_RUN.clear()
    LAST_RUN.update({"host_plan": host.as_dict(), "ranks": []})
    t0 = time.time()
    if total_callers == 1:
        _setup(0, 1, args, file_chunks, devices, plans=host.ranks)
        LAST_RUN["ranks"] = [dict(LAST_PREDICT)]
        LAST_RUN["seconds"] = round(time.time() - t0, 3)
        return _collect_streams(True, None)
    collectors = _start_collectors(output_filepath, total_callers, total_stitch_threads, num_workers, file_chunks)
    if collectors is not None:
        args = args + (collectors.export_spec(),)
    try:
        results, failed = run_ranks(_setup, [(r, total_callers, args, file_chunks, devices, host.ranks)
                                             for r in range(total_callers)],
                                    [rp.slot_prefix for rp in host.ranks])
    except BaseException:
        if collectors is not None:
            collectors.abort()
        raise
    LAST_RUN["ranks"] = [results[r] for r in sorted(results)]
    LAST_RUN["seconds"] = round(time.time() - t0, 3)
    return _streams_of_run(collectors, results, failed, total_callers, "ranks")
