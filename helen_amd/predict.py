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
"""
import multiprocessing as mp
import queue
import sys
import threading
import time

import numpy as np

from .data_store import DataStore
from .model_handler import ModelHandler
from .options import ImageSizeOptions
from .sequence_dataset import SequenceDataset, SharedSlot, fill_shared

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
    prediction_data_file = DataStore(output_filename + "_" + str(rank) + ".hdf", mode="w")
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
        print(output_filename + "_" + str(rank) + ".hdf")
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
    writer = threading.Thread(target=_writer_loop, args=(wq, prediction_data_file, free_slots, werr),
                              daemon=True)
    for k in STAGE_SECONDS:
        STAGE_SECONDS[k] = 0.0
    feeder.start()
    writer.start()
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
            bases, rles = engine.polish_host(slot.images[:n])
            STAGE_SECONDS["read_wait"] += t1 - t0
            STAGE_SECONDS["device"] += time.time() - t1
            wq.put((slot, n, bases, rles))
            batch_iterator += nb
            if rank == 0:
                eta = (time.time() - start_time) / batch_iterator * (total_batches - batch_iterator)
                sys.stderr.write("INFO: BATCHES DONE: %d/%d. ESTIMATED TIME LEFT: %d MINS %d SECS.\n"
                                 % (batch_iterator, total_batches, int(eta // 60), int(eta) % 60))
            if werr:
                break
    finally:
        wq.put(None)
        writer.join()
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        for sl in slots:
            sl.close()
    if ferr:
        raise ferr[0]
    if werr:
        raise werr[0]
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
