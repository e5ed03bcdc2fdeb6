"""Schema check of a MarginPolish image directory.

The input schema is only known from the reference's reader (models/dataloader_predict.py:64-70; labeled
files add label_base / label_run_length, models/dataloader.py:59-61).  This reports, per file, what is
actually stored -- dataset presence, type class / width, shape, layout, filters -- and flags whatever the
path cannot take, so a real file can be vetted before a run:

    python -m helen_amd check_images -i <image_dir> [--images-per-file 8]
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
