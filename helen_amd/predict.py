"""predict / predict_gpu: the per-device inference loop of `helen call_consensus`.

Same signatures and observable behaviour as helen/modules/python/models/predict_gpu.py:38-226:
one process per device, each with its own model replica and file list, writing
`<output_filename>_<rank>.hdf`.  What differs is the mechanism: the 19-chunk sliding window, the
softmax-accumulate and the argmax of a whole batch run as one helen_polish_host call on the
MI355X; several loader batches are coalesced per call (windows are independent; hidden is zeroed
per window, so coalescing cannot change results), images go up as uint8 through pinned
double-buffered copies and labels come back as uint8; the HDF5 writes run on a writer thread.
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
from .sequence_dataset import SequenceDataset

# windows per device call: scratch is ~4 MB per window, 4096 windows fill 256 CUs x 2 workgroups
DEVICE_CALL_WINDOWS = 4096


def _writer_loop(q, store, err):
    try:
        while True:
            item = q.get()
            if item is None:
                return
            batch, bases, rles = item
            for i in range(len(batch.contig)):
                store.write_prediction(batch.contig[i], batch.contig_start[i], batch.contig_end[i],
                                       batch.chunk_id[i], batch.positions[i], bases[i], rles[i],
                                       batch.filenames[i])
    except Exception as e:  # surfaced by the caller
        err.append(e)


def predict(test_file, output_filename, model_path, batch_size, num_workers, rank, device_id):
    """Run inference over the image files `test_file` (a list) on device `device_id` and write
    `<output_filename>_<rank>.hdf` (predict_gpu.py:38-179)."""
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
    transducer_model.set_capacity(min(DEVICE_CALL_WINDOWS, group * batch_size))
    engine = transducer_model.engine
    if rank == 0:
        print(output_filename + "_" + str(rank) + ".hdf")
        sys.stderr.write("INFO: MI355X HIP PATH, DEVICE " + str(device_id) + ", "
                         + str(engine.device_bytes >> 20) + " MiB OF DEVICE MEMORY HELD.\n")
        sys.stderr.write("Loading data\n")

    test_data = SequenceDataset(image_directory=None, file_list=test_file)
    total_batches = test_data.num_batches(batch_size)
    wq = queue.Queue(maxsize=4 * group)
    werr = []
    writer = threading.Thread(target=_writer_loop, args=(wq, prediction_data_file, werr), daemon=True)
    writer.start()

    def flush(pending):
        images = np.concatenate([b.images for b in pending]) if len(pending) > 1 else pending[0].images
        bases, rles = engine.polish_host(images)
        s = 0
        for b in pending:
            n = b.images.shape[0]
            wq.put((b, bases[s:s + n], rles[s:s + n]))
            s += n

    start_time = time.time()
    pending, batch_iterator = [], 0
    for batch in test_data.iter_batches(batch_size, num_workers=num_workers):
        pending.append(batch)
        batch_iterator += 1
        if len(pending) == group:
            flush(pending)
            pending = []
            if rank == 0:
                done = batch_iterator
                eta = (time.time() - start_time) / done * (total_batches - done)
                sys.stderr.write("INFO: BATCHES DONE: %d/%d. ESTIMATED TIME LEFT: %d MINS %d SECS.\n"
                                 % (done, total_batches, int(eta // 60), int(eta) % 60))
        if werr:
            raise werr[0]
    if pending:
        flush(pending)
    wq.put(None)
    writer.join()
    if werr:
        raise werr[0]
    prediction_data_file.close()
    engine.close()


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
