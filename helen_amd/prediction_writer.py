"""Writer processes of the prediction stage ($HELEN_WRITERS > 1, see helen_amd.predict).

Kept apart from helen_amd.predict so that a spawned writer imports numpy and the HDF5 binding only
(not torch): process start-up is on the critical path of short runs.
"""
import numpy as np

from .data_store import DataStore
from .sequence_dataset import attach_slot


def writer_of_region(meta, writers):
    """Writer index per window: a hash of contig_start, so that every chunk of a region -- and a
    repeat of the same (region, chunk id), which must be dropped by the one file that has it
    (DataStore.py:102-124) -- lands in the same file."""
    key = meta[:, 0].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return ((key >> np.uint64(40)) % np.uint64(writers)).astype(np.int64)


def prediction_file_name(output_filename, rank, writer=0):
    return output_filename + "_" + str(rank) + ("" if writer == 0 else "_w" + str(writer)) + ".hdf"


def writer_process(k, writers, filename, task_q, done_q):
    """Writer process k of `writers`: for every device call, store the windows of its regions.  The file is
    created when the first window arrives: a writer no region hashes to leaves NO file behind (an HDF5 file
    without a `predictions` group makes stitch -- this one and the reference's, StitchInterface.py:50-60 --
    raise)."""
    store = None
    try:
        while True:
            task = task_q.get()
            if task is None:
                break
            path, cap, n = task
            slot = attach_slot(path, cap)
            sel = np.nonzero(writer_of_region(slot.meta[:n], writers) == k)[0].astype(np.int32)
            if sel.size:
                if store is None:
                    store = DataStore(filename, mode="w")
                store.write_batch(slot.contigs[:n], slot.meta[:n], slot.positions[:n], slot.bases[:n],
                                  slot.rles[:n], sel=sel)
            done_q.put((path, None))
        if store is not None:
            store.close()
        done_q.put((None, None))
    except Exception as e:
        done_q.put((None, "writer %d: %r" % (k, e)))
