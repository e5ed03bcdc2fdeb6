"""Schema check of a MarginPolish image directory.

The input schema is only known from the reference's reader (models/dataloader_predict.py:64-70; labeled
files add label_base / label_run_length, models/dataloader.py:59-61).  This reports, per file, what is
actually stored -- dataset presence, type class / width, shape, layout, filters -- and flags whatever the
path cannot take, so a real file can be vetted before a run:

    python -m helen_amd check_images -i <image_dir> [--images-per-file 8] [--strict] [--json report.json]

`--strict` is the readiness check for REAL MarginPolish output (no such file exists offline, BASELINE.json
configs[4]): every image's schema is inspected, not a sample, and every image is then READ through the product's
own reader (the one `call_consensus` uses), recording which path took each file -- the direct scanner
(helen_amd/csrc/h5scan.h) or libhdf5 behind it -- and any refusal (`IMAGE SIZE ERROR`, a name that is too long, a
storage form neither path reads).  Exit code 0 = ready, 1 = schema problems, 2 = the reader refused something.
`--json` writes the whole report (per file: image count, dataset types / shapes / layouts / filters seen, reader
path, problems, refusals) for attaching to a bug report.
"""
import sys

from . import hdf5, native_io
from .file_manager import get_file_paths_from_directory
from .options import ImageSizeOptions

REQUIRED = ("contig", "contig_start", "contig_end", "feature_chunk_idx", "image", "position")
LABELS = ("label_base", "label_run_length")
FILTER_NAMES = {1: "deflate", 2: "shuffle", 3: "fletcher32", 4: "szip", 5: "nbit", 6: "scaleoffset",
                32000: "lzf", 32001: "blosc", 32004: "lz4", 32015: "zstd"}


def check_image(f, name):
    """-> (facts dict per dataset, list of problems) for image group `name` of open file `f`."""
    facts, problems, size_errors = {}, [], []
    L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
    base = "images/" + name + "/"
    for ds in REQUIRED + LABELS:
        if (base + ds) not in f:
            if ds in REQUIRED:
                problems.append("missing dataset '%s'" % ds)
            continue
        i = facts[ds] = f.info(base + ds)
        if not i["filters_available"]:
            problems.append("%s: filter(s) %s not available in this libhdf5" % (
                ds, [FILTER_NAMES.get(x, x) for x in i["filters"]]))
    im, po = facts.get("image"), facts.get("position")
    if im is not None:
        # the reader does np.array(image, dtype=np.uint8) (dataloader_predict.py:69): integers of any width and
        # floats are all taken (values outside 0..255 wrap there as they do here)
        if im["class"] not in ("int", "float") or len(im["shape"]) != 2 or im["shape"][1] != H:
            problems.append("image is %s %s, expected integers [l <= %d, %d]" % (im["class"], im["shape"], L, H))
            size_errors.append(tuple(im["shape"]))
        elif im["shape"][0] > L:
            problems.append("image has %d positions (> SEQ_LENGTH %d)" % (im["shape"][0], L))
            size_errors.append(tuple(im["shape"]))
    if po is not None:
        if po["class"] != "int" or len(po["shape"]) != 2 or po["shape"][1] != 3:
            problems.append("position is %s %s, expected integers [l, 3]" % (po["class"], po["shape"]))
        elif im is not None and len(im["shape"]) == 2 and po["shape"][0] != im["shape"][0]:
            problems.append("position has %d rows, image %d" % (po["shape"][0], im["shape"][0]))
            size_errors.append(tuple(im["shape"]))
    for ds in ("contig_start", "contig_end", "feature_chunk_idx"):
        i = facts.get(ds)
        if i is not None and (i["class"] != "int" or int(_count(i["shape"])) < 1):
            problems.append("%s is %s %s, expected at least one integer" % (ds, i["class"], i["shape"]))
    if "contig" in facts and facts["contig"]["class"] != "string":
        problems.append("contig is %s, expected a string" % facts["contig"]["class"])
    for ds in LABELS:
        i = facts.get(ds)
        if i is not None and im is not None and (i["class"] != "int" or i["shape"] != im["shape"][:1]):
            problems.append("%s is %s %s, expected integers [%d]" % (ds, i["class"], i["shape"], im["shape"][0]))
    facts["_size_errors"] = size_errors     # shapes the reader raises IMAGE SIZE ERROR on (dataloader_predict.py:85-86)
    return facts, problems


def _count(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _describe(i):
    if i["class"] == "int":
        t = "%sint%d" % ("" if i["signed"] else "u", 8 * i["size"])
    elif i["class"] == "float":
        t = "float%d" % (8 * i["size"])
    elif i["class"] == "string":
        t = "string" if i["variable_string"] else "string[%d]" % i["size"]
    else:
        t = i["class"]
    s = "%s %s %s" % (t, list(i["shape"]), i["layout"])
    if i["chunk"]:
        s += " chunk %s" % list(i["chunk"])
    if i["filters"]:
        s += " filters %s" % [FILTER_NAMES.get(x, x) for x in i["filters"]]
    if i["variable_string"]:
        s += " (variable-length)"
    return s


def check_image_directory(image_dir, images_per_file=8, out=sys.stdout, size_errors=None):
    """Vet every file of `image_dir`; prints a report, returns the number of problems found.  `size_errors`
    (a list) collects (file, shape) of images the reader would refuse with IMAGE SIZE ERROR."""
    import os
    n_problems = 0
    if not os.path.isdir(image_dir):
        out.write("NOT A DIRECTORY: " + str(image_dir) + "\n")
        return 1
    files = get_file_paths_from_directory(image_dir)
    if not files:
        out.write("NO .h5 / .hdf5 FILES IN " + str(image_dir) + "\n")
        return 1
    for path in files:
        with hdf5.File(path, "r") as f:
            if "images" not in f:
                out.write("%s: no 'images' group (the reader warns and skips the file)\n" % path)
                continue
            # the native listing walks the group's B-tree directly; the ctypes binding calls back per name
            names = native_io.list_images(path) if native_io.available() else f.keys("images")
            out.write("%s: %d images\n" % (path, len(names)))
            step = max(1, len(names) // max(1, images_per_file))
            shown = False
            for name in names[::step][:images_per_file]:
                facts, problems = check_image(f, name)
                if size_errors is not None:
                    size_errors.extend((path, shape) for shape in facts.pop("_size_errors"))
                else:
                    facts.pop("_size_errors")
                if not shown:
                    for ds in REQUIRED + LABELS:
                        if ds in facts:
                            out.write("    %-18s %s\n" % (ds, _describe(facts[ds])))
                    shown = True
                for p in problems:
                    out.write("  PROBLEM %s: %s\n" % (name, p))
                n_problems += len(problems)
    out.write("%d problem(s)\n" % n_problems)
    return n_problems


def _type_name(i):
    if i["class"] == "int":
        return "%sint%d" % ("" if i["signed"] else "u", 8 * i["size"])
    if i["class"] == "float":
        return "float%d" % (8 * i["size"])
    if i["class"] == "string":
        return "string" if i["variable_string"] else "string[%d]" % i["size"]
    return str(i["class"])


def read_through_the_product_reader(path, names, batch=256):
    """Read `names` of `path` through helen_amd.sequence_dataset.fill_batch -- what a reader process of predict()
    runs.  -> (images read, read by the direct scanner, read by libhdf5, [refusal strings])."""
    import numpy as np

    from .sequence_dataset import fill_batch
    L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
    images = np.empty((batch, L, H), np.uint8)
    positions = np.empty((batch, L, 3), np.int64)
    meta = np.empty((batch, 3), np.int64)
    contigs = np.zeros((batch, native_io.NAME_BYTES), np.uint8)
    refusals, done = [], 0
    before = native_io.reader_counts() if native_io.available() else (0, 0)
    for lo in range(0, len(names), batch):
        part = [(path, n) for n in names[lo:lo + batch]]
        try:
            fill_batch(part, images[:len(part)], positions[:len(part)], meta[:len(part)], contigs[:len(part)])
            done += len(part)
        except Exception:               # noqa: BLE001 -- find the image(s): one by one
            for pair in part:
                try:
                    fill_batch([pair], images[:1], positions[:1], meta[:1], contigs[:1])
                    done += 1
                except Exception as e:  # noqa: BLE001 -- every refusal is part of the report
                    refusals.append("%s: %s: %s" % (pair[1], type(e).__name__, e))
    after = native_io.reader_counts() if native_io.available() else (0, done)
    return done, after[0] - before[0], after[1] - before[1], refusals


def image_directory_report(image_dir, images_per_file=8, strict=False):
    """The report behind `check_images`: a JSON-able dict.  strict=False inspects `images_per_file` evenly spaced
    images per file; strict=True inspects every image and reads every image through the product reader."""
    import os
    report = {"image_dir": os.path.abspath(str(image_dir)), "strict": bool(strict), "files": [], "problems": 0,
              "refusals": 0, "images": 0, "images_inspected": 0, "images_read": 0,
              "expects": {"image": "integers [l <= %d, %d]" % (ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT),
                          "position": "integers [l, 3]", "contig": "string",
                          "contig_start / contig_end / feature_chunk_idx": "at least one integer",
                          "reference": "helen/modules/python/models/dataloader_predict.py:64-88"}}
    if not os.path.isdir(image_dir):
        report["error"] = "NOT A DIRECTORY"
        report["problems"] = 1
        return report
    files = get_file_paths_from_directory(image_dir)
    if not files:
        report["error"] = "NO .h5 / .hdf5 FILES"
        report["problems"] = 1
        return report
    for path in files:
        entry = {"path": path, "images": 0, "datasets": {}, "problems": [], "refusals": [], "reader_path": None,
                 "size_errors": []}
        report["files"].append(entry)
        try:
            f = hdf5.File(path, "r")
        except Exception as e:          # noqa: BLE001
            entry["refusals"].append("cannot open: %s" % e)
            report["refusals"] += 1
            continue
        with f:
            if "images" not in f:
                entry["note"] = "no 'images' group (the reader warns and skips the file)"
                continue
            names = native_io.list_images(path) if native_io.available() else f.keys("images")
            entry["images"] = len(names)
            if native_io.available() and names:
                # what the host plan of a run will price this file with (helen_amd.host_plan.READER_RATE)
                from .host_plan import READER_RATE
                entry["storage_class"] = native_io.image_storage(path)
                entry["predicted_windows_per_s_per_reader"] = READER_RATE.get(entry["storage_class"])
            report["images"] += len(names)
            step = 1 if strict else max(1, len(names) // max(1, images_per_file))
            chosen = names if strict else names[::step][:images_per_file]
            for name in chosen:
                facts, problems = check_image(f, name)
                entry["size_errors"].extend(list(s) for s in facts.pop("_size_errors"))
                for ds, i in facts.items():
                    d = entry["datasets"].setdefault(ds, {"types": [], "layouts": [], "filters": [], "chunks": [],
                                                          "rows_min": None, "rows_max": None, "describe": _describe(i)})
                    for key, val in (("types", _type_name(i)), ("layouts", i["layout"]),
                                     ("filters", [FILTER_NAMES.get(x, x) for x in i["filters"]]),
                                     ("chunks", list(i["chunk"]) if i["chunk"] else None)):
                        if val not in d[key] and val not in (None, []):
                            d[key].append(val)
                    if i["shape"]:
                        r = int(i["shape"][0])
                        d["rows_min"] = r if d["rows_min"] is None else min(d["rows_min"], r)
                        d["rows_max"] = r if d["rows_max"] is None else max(d["rows_max"], r)
                entry["problems"].extend("%s: %s" % (name, p) for p in problems)
            report["images_inspected"] += len(chosen)
        if strict and entry["images"]:
            done, fast, lib, refusals = read_through_the_product_reader(path, names)
            entry["images_read"] = done
            entry["reader_path"] = ("direct scanner" if lib == 0 else "libhdf5" if fast == 0 else
                                    "mixed (%d direct, %d libhdf5)" % (fast, lib))
            entry["refusals"].extend(refusals)
            report["images_read"] += done
        report["problems"] += len(entry["problems"])
        report["refusals"] += len(entry["refusals"])
    report["verdict"] = ("the reader refuses some images" if report["refusals"] else
                         "schema problems" if report["problems"] else
                         "ready" if strict else "sample looks fine (run with --strict for the full check)")
    report["exit_code"] = 2 if report["refusals"] else 1 if report["problems"] else 0
    return report


def main(image_dir, images_per_file=8, strict=False, json_path=None, out=sys.stdout):
    """`python -m helen_amd check_images`: the printed report (as before), plus --strict / --json."""
    import json
    if not strict and not json_path:
        return 1 if check_image_directory(image_dir, images_per_file, out=out) else 0
    rep = image_directory_report(image_dir, images_per_file, strict)
    for e in rep["files"]:
        out.write("%s: %d images%s%s\n" % (e["path"], e["images"],
                                           (", read through the " + e["reader_path"]) if e.get("reader_path") else "",
                                           (", stored %s: ~%d windows/s per reader" % (e["storage_class"],
                                                                                    e["predicted_windows_per_s_per_reader"]))
                                           if e.get("storage_class") else ""))
        for ds in REQUIRED + LABELS:
            if ds in e["datasets"]:
                d = e["datasets"][ds]
                rows = "" if d["rows_min"] is None else " rows %s..%s" % (d["rows_min"], d["rows_max"])
                out.write("    %-18s %s %s%s%s\n" % (ds, "/".join(d["types"]), "/".join(d["layouts"]), rows,
                                                     (" filters %s" % d["filters"]) if d["filters"] else ""))
        for p in e["problems"]:
            out.write("  PROBLEM %s\n" % p)
        for p in e["refusals"]:
            out.write("  REFUSED %s\n" % p)
    if "error" in rep:
        out.write(rep["error"] + ": " + rep["image_dir"] + "\n")
    out.write("%d image(s), %d inspected, %d read through the product reader; %d problem(s), %d refusal(s): %s\n"
              % (rep["images"], rep["images_inspected"], rep["images_read"], rep["problems"], rep["refusals"],
                 rep.get("verdict", rep.get("error"))))
    if json_path:
        text = json.dumps(rep, indent=1, sort_keys=True)
        if json_path == "-":
            out.write(text + "\n")
        else:
            with open(json_path, "w") as fh:
                fh.write(text + "\n")
    return rep.get("exit_code", 1)
