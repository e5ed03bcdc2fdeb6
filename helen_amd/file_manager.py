"""Directory / file-list helpers of the polish path (reference: helen/modules/python/FileManager.py:5-70
and CallConsensusInterface.py:36-44)."""
import os


def handle_output_directory(output_dir):
    """Create `output_dir` if needed and return its absolute path with a trailing separator
    (FileManager.py:10-26)."""
    output_dir = os.path.abspath(output_dir)
    os.makedirs(output_dir, exist_ok=True)
    return output_dir if output_dir.endswith(os.sep) else output_dir + os.sep


def get_file_paths_from_directory(directory_path, sort=True):
    """Absolute paths of the `*h5` files of a directory (CallConsensusInterface.py:36-44 tests the
    last two characters of the name).  The reference uses raw os.listdir order;
    we sort for run-to-run determinism -- pass sort=False for the raw order."""
    names = os.listdir(directory_path)
    if sort:
        names = sorted(names)
    return [os.path.abspath(os.path.join(directory_path, f)) for f in names
            if os.path.isfile(os.path.join(directory_path, f)) and f[-2:] == "h5"]


def chunk_it(seq, num):
    """Split `seq` into `num` contiguous runs of near-equal length (FileManager.py:52-70)."""
    avg = len(seq) / float(num)
    out = []
    last = 0.0
    while last < len(seq):
        out.append(seq[int(last):int(last + avg)])
        last += avg
    return out


def shard_round_robin(files, callers):
    """File-level sharding exactly as CallConsensusInterface.py:138-145: file i goes to caller
    i % callers; empty shards are dropped."""
    chunks = [[] for _ in range(callers)]
    for i, f in enumerate(files):
        chunks[i % callers].append(f)
    return [c for c in chunks if len(c) > 0]
