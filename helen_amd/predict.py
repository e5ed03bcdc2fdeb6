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

$HELEN_WRITERS=W (default 1) shards the prediction writer over W processes per rank: creating the
three small HDF5 datasets of a window costs ~70 us inside libhdf5, ~14 k windows/s per process.
Writer 0 keeps the reference's file name `<output>_<rank>.hdf`, writer k > 0 writes
`<output>_<rank>_w<k>.hdf`; all chunks of one region go to the same file, and stitch takes every
`*.hdf` of the directory (StitchInterface.py:35-36), so the result is the same.
"""
import multiprocessing as mp
import os
import queue
import sys
import threading
import time

import numpy as np

from .data_store import DataStore
from .model_handler import ModelHandler
from .options import ImageSizeOptions
from .sequence_dataset import SequenceDataset, SharedSlot, attach_slot, fill_shared

# windows per device call: scratch is ~4 MB per window, 4096 windows fill 256 CUs x 2 workgroups
DEVICE_CALL_WINDOWS = 4096

# wall seconds per pipeline stage of the last predict() in this process (reported by rank 0)
STAGE_SECONDS = {"read_wait": 0.0, "device": 0.0, "write": 0.0}


def _writer_loop(wq, store, free_slots, err):
    """Writer thread: labels of one device call -> prediction HDF5, then recycle the slot."""
    try:
        while True:
            item = wq.get()
            if item is None:
                return
            slot, n, bases, rles = item
            t0 = time.time()
            store.write_batch(slot.contigs[:n], slot.meta[:n], slot.positions[:n], bases, rles)
            STAGE_SECONDS["write"] += time.time() - t0
            free_slots.put(slot)
    except Exception as e:  # surfaced by the caller
        err.append(e)
        free_slots.put(None)


def writer_of_region(meta, writers):
    """Writer index per window: a hash of contig_start, so that every chunk of a region -- and a
    repeat of the same (region, chunk id), which must be dropped by the one file that has it
    (DataStore.py:102-124) -- lands in the same file."""
    key = meta[:, 0].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return ((key >> np.uint64(40)) % np.uint64(writers)).astype(np.int64)


def prediction_file_name(output_filename, rank, writer=0):
    return output_filename + "_" + str(rank) + ("" if writer == 0 else "_w" + str(writer)) + ".hdf"


def _writer_process(k, writers, filename, task_q, done_q):
    """Writer process k of `writers`: for every device call, store the windows of its regions."""
    try:
        store = DataStore(filename, mode="w")
        while True:
            task = task_q.get()
            if task is None:
                break
            path, cap, n = task
            slot = attach_slot(path, cap)
            sel = np.nonzero(writer_of_region(slot.meta[:n], writers) == k)[0].astype(np.int32)
            if sel.size:
                store.write_batch(slot.contigs[:n], slot.meta[:n], slot.positions[:n], slot.bases[:n],
                                  slot.rles[:n], sel=sel)
            done_q.put((path, None))
        store.close()
        done_q.put((None, None))
    except Exception as e:
        done_q.put((None, "writer %d: %r" % (k, e)))


class _WriterPool(object):
    """W writer processes; a slot is recycled when all of them are done with it."""

    def __init__(self, output_filename, rank, writers, free_slots, err):
        ctx = mp.get_context("spawn")
        self.writers, self.free_slots, self.err = writers, free_slots, err
        self.done_q = ctx.Queue()
        self.task_qs = [ctx.Queue() for _ in range(writers)]
        self.procs = [ctx.Process(target=_writer_process, daemon=True,
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


def _feeder_loop(calls, free_slots, ready_q, pool, cap, err):
    """Feeder thread: for each device call take a free slot and get its loader batches read into
    it -- by the worker pool (one task per loader batch) or inline when num_workers == 0."""
    try:
        for batches in calls:
            slot = free_slots.get()
            if slot is None:
                return
            futures, off = [], 0
            for pairs in batches:
                if pool is not None:
                    futures.append(pool.submit(fill_shared, slot.path, cap, off, pairs))
                else:
                    fill_shared(slot.path, cap, off, pairs)
                off += len(pairs)
            ready_q.put((slot, off, futures, len(batches)))
        ready_q.put(None)
    except Exception as e:
        err.append(e)
        ready_q.put(None)


def predict(test_file, output_filename, model_path, batch_size, num_workers, rank, device_id):
    """Run inference over the image files `test_file` (a list) on device `device_id` and write
    `<output_filename>_<rank>.hdf` (predict_gpu.py:38-179).

    Pipeline (three stages, three shared-memory slots of one device call each):
      reader processes fill slot k+1 | the GPU polishes slot k | the writer thread stores slot k-1."""
    import torch
    writers = max(1, int(os.environ.get("HELEN_WRITERS", "1")))
    prediction_data_file = DataStore(prediction_file_name(output_filename, rank), mode="w") \
        if writers == 1 else None
    transducer_model, hidden_size, gru_layers, prev_ite = ModelHandler.load_simple_model(
        model_path, input_channels=ImageSizeOptions.IMAGE_CHANNELS,
        image_features=ImageSizeOptions.IMAGE_HEIGHT, seq_len=ImageSizeOptions.SEQ_LENGTH,
        num_base_classes=ImageSizeOptions.TOTAL_BASE_LABELS,
        num_rle_classes=ImageSizeOptions.TOTAL_RLE_LABELS)
    transducer_model.eval()
    torch.cuda.set_device(device_id)
    transducer_model.to(device_id)
    group = max(1, DEVICE_CALL_WINDOWS // batch_size)       # loader batches per device call
    cap = group * batch_size
    transducer_model.set_capacity(min(DEVICE_CALL_WINDOWS, cap))
    engine = transducer_model.engine
    if rank == 0:
        print(prediction_file_name(output_filename, rank))
        sys.stderr.write("INFO: MI355X HIP PATH, DEVICE " + str(device_id) + ", "
                         + str(engine.device_bytes >> 20) + " MiB OF DEVICE MEMORY HELD.\n")
        sys.stderr.write("Loading data\n")

    test_data = SequenceDataset(image_directory=None, file_list=test_file)
    pairs = test_data.all_images
    batches = [pairs[i:i + batch_size] for i in range(0, len(pairs), batch_size)]   # sequential,
    calls = [batches[i:i + group] for i in range(0, len(batches), group)]           # short last batch
    total_batches = len(batches)

    slots = [SharedSlot(cap) for _ in range(min(3, max(1, len(calls))))]
    free_slots, ready_q, wq = queue.Queue(), queue.Queue(maxsize=2), queue.Queue()
    for sl in slots:
        free_slots.put(sl)
    pool = None
    if num_workers > 0 and calls:
        import concurrent.futures as cf
        pool = cf.ProcessPoolExecutor(num_workers, mp_context=mp.get_context("spawn"))
    ferr, werr = [], []
    feeder = threading.Thread(target=_feeder_loop, args=(calls, free_slots, ready_q, pool, cap, ferr),
                              daemon=True)
    for k in STAGE_SECONDS:
        STAGE_SECONDS[k] = 0.0
    if writers == 1:
        writer = threading.Thread(target=_writer_loop, args=(wq, prediction_data_file, free_slots, werr),
                                  daemon=True)
        writer.start()
        writer_pool = None
    else:
        writer_pool = _WriterPool(output_filename, rank, writers, free_slots, werr)
    feeder.start()
    start_time = time.time()
    batch_iterator = 0
    try:
        while True:
            t0 = time.time()
            item = ready_q.get()
            if item is None:
                break
            slot, n, futures, nb = item
            for f in futures:
                f.result()                       # raises the reader's exception, if any
            t1 = time.time()
            bases, rles = engine.polish_host(slot.images[:n], out=(slot.bases[:n], slot.rles[:n]))
            STAGE_SECONDS["read_wait"] += t1 - t0
            STAGE_SECONDS["device"] += time.time() - t1
            if writer_pool is None:
                wq.put((slot, n, bases, rles))
            else:
                writer_pool.submit(slot, n)
            batch_iterator += nb
            if rank == 0:
                eta = (time.time() - start_time) / batch_iterator * (total_batches - batch_iterator)
                sys.stderr.write("INFO: BATCHES DONE: %d/%d. ESTIMATED TIME LEFT: %d MINS %d SECS.\n"
                                 % (batch_iterator, total_batches, int(eta // 60), int(eta) % 60))
            if werr:
                break
    finally:
        if writer_pool is None:
            wq.put(None)
            writer.join()
        else:
            writer_pool.close()
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        for sl in slots:
            sl.close()
    if ferr:
        raise ferr[0]
    if werr:
        raise werr[0]
    if prediction_data_file is not None:
        prediction_data_file.close()
    engine.close()
    if rank == 0:
        sys.stderr.write("INFO: %d WINDOWS IN %.1f SECS (WAITING FOR READERS %.1f, DEVICE %.1f, WRITER BUSY %.1f).\n"
                         % (len(pairs), time.time() - start_time, STAGE_SECONDS["read_wait"],
                            STAGE_SECONDS["device"], STAGE_SECONDS["write"]))


def _setup(rank, total_callers, args, all_input_files, all_devices):
    output_filepath, model_path, batch_size, num_workers = args
    predict(all_input_files[rank], output_filepath, model_path, batch_size, num_workers, rank,
            all_devices[rank])


def predict_gpu(file_chunks, output_filepath, model_path, batch_size, total_callers, devices,
                num_workers):
    """One process per device, each over its own file list (predict_gpu.py:207-226).  A failing
    child raises here, like mp.spawn(join=True) does."""
    args = (output_filepath, model_path, batch_size, num_workers)
    if total_callers == 1:
        _setup(0, 1, args, file_chunks, devices)
        return
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_setup, args=(r, total_callers, args, file_chunks, devices))
             for r in range(total_callers)]
    for p in procs:
        p.start()
    failed = []
    for r, p in enumerate(procs):
        p.join()
        if p.exitcode != 0:
            failed.append((r, p.exitcode))
    if failed:
        raise RuntimeError("prediction process(es) failed: " + ", ".join(
            "rank %d exit %s" % f for f in failed))
