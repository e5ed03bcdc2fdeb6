"""call_consensus: argument validation, device planning, file-level sharding and dispatch
(helen/modules/python/CallConsensusInterface.py:47-156), then `polish_genome`
(PolishInterface.py:49-105) on top of it."""
import os
import sys
import time
from datetime import datetime

from . import file_manager
from .predict import predict_gpu


def _err(msg):
    sys.stderr.write("ERROR: " + msg + "\n")


def plan_devices(device_ids, visible_devices):
    """-> (device list, callers).  None = all visible devices (CallConsensusInterface.py:107-112);
    otherwise a comma-separated id string (:113-126)."""
    if device_ids is None:
        devs = list(range(visible_devices))
    else:
        devs = [int(i) for i in str(device_ids).split(",") if i != ""]
        for d in devs:
            if d < 0 or d >= visible_devices:
                raise ValueError("GPU DEVICE: %d IS NOT AVAILABLE (VISIBLE: %d)" % (d, visible_devices))
    return devs, len(devs)


def vet_image_directory(image_dir, images_per_file=4):
    """Schema check of a few images of every file before any process is started (helen_amd.check_images,
    SURVEY.md 8f-2): the report goes to stderr only when something is off, and an image the reader would refuse
    raises its error right here (`IMAGE SIZE ERROR`, dataloader_predict.py:85-86) instead of from a worker
    process minutes into the run.  Returns the number of (non-fatal) findings."""
    import io

    from .check_images import check_image_directory
    report, size_errors = io.StringIO(), []
    n = check_image_directory(image_dir, images_per_file=images_per_file, out=report, size_errors=size_errors)
    if n:
        sys.stderr.write("WARN: IMAGE DIRECTORY CHECK (python -m helen_amd check_images -i %s):\n%s"
                         % (image_dir, report.getvalue()))
    if size_errors:
        path, shape = size_errors[0]
        raise ValueError("IMAGE SIZE ERROR: " + str(path) + " " + str(tuple(shape)))
    return n


def call_consensus(image_dir, model_path, batch_size, num_workers, threads, output_dir,
                   output_prefix, gpu_mode, device_ids, callers, stitch_threads=None):
    """(CallConsensusInterface.py:47-156.)  `stitch_threads` is polish_genome's: the regions are decoded and their overlaps
    aligned behind the inference (helen_amd.stitch_stream) and come back as the return value (None otherwise)."""
    if not os.path.isfile(model_path):
        _err("CAN NOT LOCATE MODEL FILE.")
        sys.exit(1)
    if not os.path.isdir(image_dir):
        _err("CAN NOT LOCATE IMAGE DIRECTORY.")
        sys.exit(1)
    if batch_size <= 0:
        _err("batch_size NEEDS TO BE >0.")
        sys.exit(1)
    if num_workers < 0:
        _err("num_workers NEEDS TO BE >=0.")
        sys.exit(1)
    if threads <= 0:
        _err("THREAD NEEDS TO BE >=0.")
        sys.exit(1)
    output_dir = file_manager.handle_output_directory(output_dir)
    output_filename = os.path.join(output_dir, output_prefix)
    sys.stderr.write("INFO: OUTPUT FILE: " + output_filename + "\n")

    if gpu_mode:
        # how many devices: from the library itself when this run stays clear of torch (helen_amd.predict decides the same
        # way, from the same two facts), else from torch
        from .predict import native_model_state, native_path_wanted
        try:
            torch_free = native_path_wanted() and native_model_state(model_path) is not None
        except (ValueError, RuntimeError) as e:       # a checkpoint of another architecture: what the loader would say
            _err(str(e))
            sys.exit(1)
        if torch_free:
            from .native_engine import device_count
            visible = device_count()
        else:
            import torch
            visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if visible <= 0:
            # (never a silent switch to the host path: --gpu_mode means the MI355X)
            _err("NO MI355X VISIBLE (NO HIP DEVICE FOR THIS PROCESS).")
            sys.exit(1)
        try:
            device_ids, callers = plan_devices(device_ids, visible)
        except ValueError as e:
            _err(str(e))
            sys.exit(1)
        sys.stderr.write("INFO: AVAILABLE GPU DEVICES: " + str(device_ids) + "\n")
    else:
        # the reference's CPU mode (CallConsensusInterface.py:128-131,150-153): `callers` processes of threads // callers
        # threads each -- here around libhelen_cpu.so (helen_amd/csrc/cpu_path.cpp) instead of an ONNX Runtime session
        if callers <= 0:
            _err("callers NEEDS TO BE >0.")
            sys.exit(1)
        threads_per_caller = max(1, int(threads / callers))
        sys.stderr.write("INFO: HOST PATH: %d CALLER(S) OF %d THREAD(S).\n" % (callers, threads_per_caller))

    input_files = file_manager.get_file_paths_from_directory(image_dir)
    file_chunks = file_manager.shard_round_robin(input_files, callers)
    callers = len(file_chunks)
    if callers == 0:
        _err("NO IMAGE FILES (*.h5) FOUND IN " + image_dir)
        sys.exit(1)
    vet = None
    if os.environ.get("HELEN_SKIP_IMAGE_CHECK", "") != "1":
        # The schema check of a few images per file runs BESIDE the prediction, whatever the number of ranks (the reader
        # raises the same IMAGE SIZE ERROR itself when it meets such an image; the check's findings are reported either
        # way).  Listing the images of a whole-genome directory walks every file's group structures -- 4.6 s of one CPU
        # and 14 KB of mapped file pages per window on 3 M deflated windows (profiles/r06_genome_scale_one_device.txt) --
        # which until round 6 a multi-rank run spent before its first process started.  With one rank in this process the
        # check enters libhdf5 from its own thread (helen_amd.hdf5, ctypes): it holds the lock libhelen_io.so serialises
        # its own library calls with, so a libhdf5 that is not built thread-safe is never entered from two threads (a
        # reader or writer that needs the library waits for the check to finish; the direct scanner and emitter never do).
        import threading

        from . import native_io
        vet = {"error": None}

        def run_vet():
            try:
                with native_io.library_lock():
                    vet_image_directory(image_dir)
            except BaseException as e:          # noqa: BLE001 -- re-raised below
                vet["error"] = e
        vet["thread"] = threading.Thread(target=run_vet, daemon=True)
        vet["thread"].start()
    streams = None
    try:
        if gpu_mode:
            streams = predict_gpu(file_chunks, output_filename, model_path, batch_size, callers, device_ids,
                                  num_workers, stitch_threads=stitch_threads)
        else:
            from .predict import predict_cpu
            streams = predict_cpu(file_chunks, output_filename, model_path, batch_size, callers, threads_per_caller,
                                  num_workers, stitch_threads=stitch_threads)
    finally:
        if vet is not None:
            vet["thread"].join()
    if vet is not None and vet["error"] is not None:
        if hasattr(streams, "abort"):
            streams.abort()
        raise vet["error"]
    sys.stderr.write("INFO: PREDICTION GENERATED SUCCESSFULLY.\n")
    return streams


def _process_age():
    """Seconds since this process was started (fork + exec, interpreter start-up and imports included), from
    /proc/self/stat's start time and /proc/uptime; None where /proc does not say."""
    try:
        with open("/proc/self/stat") as f:
            start_ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / float(os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return None


def _report_ranks(run):
    """Several ranks: one line per rank with its stage times (rank 0 alone prints the long line while it runs), and the
    resident-memory high-water marks of the parent, the ranks and the stitch collectors."""
    from .host_plan import peak_rss_mb, rss_breakdown_mb
    ranks = run.get("ranks") or []
    if len(ranks) > 1:
        for r in ranks:
            st = r.get("stage_seconds") or {}
            sys.stderr.write("INFO: RANK %s: %s WINDOWS IN %.1f SECS (WAITING FOR READERS %.1f, DEVICE %.1f, WRITER BUSY %.1f, "
                             "STITCH STAGE BUSY %.1f), %s READER(S).\n"
                             % (r.get("rank"), r.get("windows"), r.get("seconds") or 0.0, st.get("read_wait", 0.0), st.get("device", 0.0),
                                st.get("write", 0.0), st.get("stitch", 0.0), r.get("reader_workers")))
    collectors = (run.get("stitch_collectors") or {}).get("per_collector") or []
    # VmHWM, and beside it the anonymous part at the end of each process's work: the rest is pages of mapped image files
    # (the direct scanner) and of page-locked slots, resident without being heap
    sys.stderr.write("INFO: PEAK RESIDENT MEMORY (MB; IN BRACKETS THE ANONYMOUS PART AT THE END): PARENT %s (%s), RANKS %s%s.\n"
                     % (peak_rss_mb(), rss_breakdown_mb()[0],
                        ", ".join("%s (%s)" % (r.get("peak_rss_mb"), r.get("rss_anon_mb")) for r in ranks),
                        ", STITCH COLLECTORS " + ", ".join("%s (%s)" % (c.get("peak_rss_mb"), c.get("rss_anon_mb")) for c in collectors)
                        if collectors else ""))


def polish_genome(image_dir, model_path, batch_size, num_workers, threads, output_dir,
                  output_prefix, gpu_mode, device_ids, callers):
    """call_consensus into `<output_dir>/predictions_<timestamp>/`, then stitch the predictions into
    `<output_dir>/<output_prefix>.fa` (PolishInterface.py:49-105)."""
    from . import stitch_stream
    from .stitch import perform_stitch
    output_dir = file_manager.handle_output_directory(output_dir)
    timestr = datetime.now().strftime("%m%d%Y_%H%M%S")
    prediction_dir = file_manager.handle_output_directory(
        os.path.join(output_dir, "predictions_" + timestr))
    t0 = time.time()
    sys.stderr.write("INFO: RUN-ID: " + timestr + "\n")
    sys.stderr.write("INFO: PREDICTION OUTPUT DIRECTORY: " + prediction_dir + "\n")
    sys.stderr.write("INFO: CALL CONSENSUS STARTING\n")
    # stitch runs BEHIND the inference: regions are decoded and neighbours aligned while the device stage works
    # (helen_amd.stitch_stream); what follows the last window is the joins in the reference's order
    streams = call_consensus(image_dir, model_path, batch_size, num_workers, threads, prediction_dir,
                             output_prefix, gpu_mode, device_ids, callers,
                             stitch_threads=threads if stitch_stream.enabled() else None)
    t1 = time.time()
    sys.stderr.write("INFO: STITCH STARTING\n")
    print(prediction_dir)
    from . import predict as _predict
    if streams is not None:
        # the pipelined stitch is an optimisation of this command, never a way to lose its result: the prediction files
        # are complete at this point, and whatever goes wrong with the regions parked beside the run (a collector died,
        # the spill directory filled up) the FASTA comes from them in a second phase, as `helen stitch` would make it
        try:
            if hasattr(streams, "export_spec"):
                # several ranks: collector processes hold the regions and have joined them (helen_amd.stitch_collect)
                streams.finish(output_dir, output_prefix)
                _predict.LAST_RUN["stitch_collectors"] = streams.stats      # (what each collector did, for reports)
            else:
                stitch_stream.finish_stitch(streams, prediction_dir, output_dir, output_prefix, threads)
        except Exception as e:          # noqa: BLE001 -- reported; the second phase takes over
            sys.stderr.write("WARNING: THE PIPELINED STITCH DID NOT FINISH (%s: %s): STITCHING THE PREDICTION FILES INSTEAD.\n"
                             % (type(e).__name__, e))
            if hasattr(streams, "abort"):
                streams.abort()
            streams = None
    if streams is None:
        perform_stitch(prediction_dir, output_dir, output_prefix, threads)
    t2 = time.time()

    def fmt(a, b):
        return "%d HOURS %d MINS %d SECS" % (int((b - a) // 3600), int((b - a) % 3600 // 60), int(b - a) % 60)
    sys.stderr.write("INFO: FINISHED PROCESSING.\n")
    age = _process_age()
    sys.stderr.write("INFO: WALL CLOCK: %sBEFORE CALL CONSENSUS %.2f S, CALL CONSENSUS %.2f S, STITCH AFTER THE LAST WINDOW %.2f S.\n"
                     % ("" if age is None else "%.2f S SINCE THE PROCESS STARTED: " % age,
                        0.0 if age is None else max(0.0, age - (t2 - t0)), t1 - t0, t2 - t1))
    _report_ranks(_predict.LAST_RUN)
    sys.stderr.write("INFO: TOTAL TIME ELAPSED: " + fmt(t0, t2) + "\n")
    sys.stderr.write("INFO: PREDICTION TIME: " + fmt(t0, t1) + "\n")
    sys.stderr.write("INFO: STITCH TIME: " + fmt(t1, t2) + "\n")
    return prediction_dir
