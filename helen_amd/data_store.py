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

from . import hdf5


class DataStore(object):
    _prediction_path_ = "predictions"

    def __init__(self, filename, mode="r"):
        self.filename = filename
        self.mode = mode
        self.file_handler = hdf5.File(filename, mode)
        self._predictions = set()
        self._predictions_contig = set()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()

    def close(self):
        self.file_handler.close()

    def write_prediction(self, contig, contig_start, contig_end, chunk_id, position,
                         predicted_bases, predicted_rles, filename=None):
        contig_start = int(contig_start)
        contig_end = int(contig_end)
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
