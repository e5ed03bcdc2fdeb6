"""TEST INFRASTRUCTURE, build container only -- what the image lacks for IMPORTING AND RUNNING modules of the reference
(`/root/reference/helen/modules/python/...`) when golden fixtures are generated (tests/golden/make_golden_stitch.py,
make_golden_io.py).  Nothing here restates reference logic; it supplies the third-party / generated modules the
reference imports:

  * `helen.build.HELEN` -- the reference's pybind11 module around its vendored striped Smith-Waterman
    (modules/headers/pybind_api.h:16-47): `HELEN.Aligner / Filter / Alignment` are thin Python classes over
    oracle/_ref/libssw_ref.so, the REFERENCE's own ssw.c / ssw_cpp.cpp compiled in place (oracle/Makefile `ref`), with the
    attribute names the binding gives them.  The alignments are the reference library's, not this package's.
  * `h5py` (not installed) -- a veneer with the handful of h5py calls the reference makes (File(path, mode) as a context
    manager or not, `name in node`, node[name], node.keys(), dataset[()], file[path] = value, file.close()) on top of
    libhdf5 itself (helen_amd/hdf5.py, the ctypes binding): the bytes come from / go through the HDF5 library, only the
    Python spelling of the calls is h5py's.  Like h5py it hands string datasets back as numpy byte strings, stores a
    Python int as a scalar int64 dataset and an array with the array's own dtype.
  * `torchvision.transforms` (not installed) -- dataloader_predict.py builds `Compose([ToTensor()])` in its constructor and
    never uses it: two empty callables.
  * `np.int`, `np.str` -- removed from numpy; aliased to the builtins they used to name.
"""
import ctypes
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_SSW = os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so")


def install():
    lib = ctypes.CDLL(REF_SSW)

    class Alignment(object):
        def __init__(self):
            self.Clear()

        def Clear(self):
            self.best_score = 0
            self.best_score2 = 0
            self.reference_begin = 0
            self.reference_end = 0
            self.query_begin = 0
            self.query_end = 0
            self.ref_end_next_best = 0
            self.mismatches = 0
            self.cigar_string = ""
            self.cigar = []

    class Filter(object):
        def __init__(self, *a):
            self.report_begin_position = True
            self.report_cigar = True
            self.score_filter = 0
            self.distance_filter = 32767

    class Aligner(object):
        def __init__(self, match=2, mismatch=2, gap_open=3, gap_extend=1):
            self.p = (match, mismatch, gap_open, gap_extend)
            self.ref = b""

        def SetReferenceSequence(self, seq, length):
            self.ref = seq.encode()[:length]
            return length

        def Align_cpp(self, query, flt, alignment, mask_len):
            alignment.Clear()
            if not self.ref or not query:
                return False
            out = (ctypes.c_int * 6)()
            cig = ctypes.create_string_buffer(16 * (len(self.ref) + len(query)) + 64)
            rc = lib.ssw_ref_align(self.ref, len(self.ref), query.encode(), *self.p, out, cig, len(cig))
            (alignment.best_score, alignment.reference_begin, alignment.reference_end, alignment.query_begin,
             alignment.query_end, alignment.mismatches) = list(out)
            alignment.cigar_string = cig.value.decode()
            return rc == 0

    helen_build = types.ModuleType("helen.build")
    helen_build.HELEN = types.SimpleNamespace(Aligner=Aligner, Filter=Filter, Alignment=Alignment)
    sys.modules["helen.build"] = helen_build

    # h5py's spelling of the few calls Stitch.py makes, on libhdf5 through helen_amd/hdf5.py
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5

    import numpy as np

    class Node(object):
        def __init__(self, f, path):
            self.f, self.path = f, path

        def _child(self, name):
            return (self.path.rstrip("/") + "/" + name) if self.path else name

        def __contains__(self, name):
            return self.f.exists(self._child(name))

        def keys(self):
            return self.f.keys(self.path or "/")

        def __getitem__(self, name):
            if name == ():                                    # dataset[()]
                value = self.f.read(self.path)
                if isinstance(value, np.ndarray) and value.dtype == object:      # strings: h5py gives byte strings
                    value = np.array([v.encode() if isinstance(v, str) else v for v in value.ravel()]).reshape(value.shape)
                return value
            return Node(self.f, self._child(name))

        def __setitem__(self, name, value):                   # file[path] = value
            if isinstance(value, (int, np.integer)) and not isinstance(value, np.ndarray):
                self.f.write(self._child(name), np.int64(value))
            else:
                value = np.asarray(value)
                self.f.write(self._child(name), value, value.dtype.type)

    class File(Node):
        def __init__(self, path, mode="r"):
            assert mode in ("r", "w")
            Node.__init__(self, hdf5.File(path, mode), "")

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.f.close()

        def close(self):
            self.f.close()

    h5py = types.ModuleType("h5py")
    h5py.File = File
    sys.modules["h5py"] = h5py
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Compose = lambda steps: (lambda x: x)
    tv.transforms.ToTensor = lambda: (lambda x: x)
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "str"):
        np.str = str
    return Alignment
