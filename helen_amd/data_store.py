"""DataStore: the prediction HDF5 file consumed by `helen stitch`.

Layout and dtypes follow helen/modules/python/DataStore.py:83-133 exactly:
    predictions/<contig>/<contig>-<start>-<end>/contig_start      scalar int64
    predictions/<contig>/<contig>-<start>-<end>/contig_end        scalar int64
    predictions/<contig>/<contig>-<start>-<end>/<chunk_id>/position  uint32 [1000, 3]
    predictions/<contig>/<contig>-<start>-<end>/<chunk_id>/bases     uint8  [1000]
    predictions/<contig>/<contig>-<start>-<end>/<chunk_id>/rles      uint8  [1000]
The -1 rows that pad short images wrap to 4294967295 in `position` (np.array(..., dtype=np.uint32)
in the reference); an image whose (contig, prefix, chunk id) was already written is silently
skipped (DataStore.py:102-124).
"""
import numpy as np

from . import hdf5, native_io


class DataStore(object):
    _prediction_path_ = "predictions"

    def __init__(self, filename, mode="r"):
        self.filename = filename
        self.mode = mode
        # writing goes through libhelen_io.so when it is built (same layout, ~2x the windows/s)
        self._native = native_io.Writer(filename) if (mode == "w" and native_io.available()) else None
        self.file_handler = None if self._native else hdf5.File(filename, mode)
        self._predictions = set()
        self._predictions_contig = set()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()

    def close(self):
        if self._native is not None:
            self._native.close()
        elif self.file_handler is not None:
            self.file_handler.close()

    def write_batch(self, contigs, meta, positions, bases, rles, sel=None):
        """write_prediction for a whole batch: contigs = list of str or packed u8 [n,256], meta i64
        [n,3] = (contig_start, contig_end, chunk_id), positions i64 [n,1000,3], labels u8 [n,1000];
        `sel` = optional row indices (in order) to write instead of all rows."""
        if self._native is not None:
            if isinstance(contigs, (list, tuple)):
                contigs = native_io.pack_contigs(contigs)
            self._native.write(contigs, meta, positions, bases, rles, sel)
            return
        names = contigs if isinstance(contigs, (list, tuple)) else native_io.contig_names(contigs)
        for i in (range(len(names)) if sel is None else [int(k) for k in sel]):
            self.write_prediction(names[i], meta[i, 0], meta[i, 1], meta[i, 2], positions[i], bases[i], rles[i])

    def write_prediction(self, contig, contig_start, contig_end, chunk_id, position,
                         predicted_bases, predicted_rles, filename=None):
        contig_start = int(contig_start)
        contig_end = int(contig_end)
        if self._native is not None:
            self._native.write(native_io.pack_contigs([str(contig)]),
                               np.array([[contig_start, contig_end, int(chunk_id)]], np.int64),
                               np.asarray(position, np.int64)[None], np.asarray(predicted_bases, np.uint8)[None],
                               np.asarray(predicted_rles, np.uint8)[None])
            return
        chunk_name_prefix = str(contig) + "-" + str(contig_start) + "-" + str(contig_end)
        chunk_name_suffix = str(int(chunk_id))
        name = contig + chunk_name_prefix + chunk_name_suffix
        root = "{}/{}/{}".format(self._prediction_path_, contig, chunk_name_prefix)
        if chunk_name_prefix not in self._predictions_contig:
            self._predictions_contig.add(chunk_name_prefix)
            self.file_handler.write(root + "/contig_start", contig_start)
            self.file_handler.write(root + "/contig_end", contig_end)
        if name not in self._predictions:
            self._predictions.add(name)
            pos = np.asarray(position).astype(np.int64).astype(np.uint32)   # -1 -> 4294967295
            self.file_handler.write(root + "/" + chunk_name_suffix + "/position", pos, np.uint32)
            self.file_handler.write(root + "/" + chunk_name_suffix + "/bases",
                                    np.asarray(predicted_bases), np.uint8)
            self.file_handler.write(root + "/" + chunk_name_suffix + "/rles",
                                    np.asarray(predicted_rles), np.uint8)
