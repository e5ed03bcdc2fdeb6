"""`helen` command line for the inference path: `polish`, `call_consensus`, `stitch`, `version`,
`torch_stat`, flag-compatible with helen/helen.py:12-222 (-i -m -b -w -t -o -p -g -d_ids -c)."""
import argparse
import sys

from . import __version__


def add_polish_arguments(parser, threads_default):
    parser.add_argument("-i", "--image_dir", type=str, required=True,
                        help="[REQUIRED] Path to a directory where all MarginPolish generated images are.")
    parser.add_argument("-m", "--model_path", type=str, required=True,
                        help="[REQUIRED] Path to a trained model (pkl file).")
    parser.add_argument("-b", "--batch_size", type=int, required=False, default=512,
                        help="Batch size for testing, default is 512.")
    parser.add_argument("-w", "--num_workers", type=int, required=False, default=8,
                        help="Number of workers to assign to the dataloader. Default is 8.")
    parser.add_argument("-t", "--threads", type=int, required=False, default=threads_default,
                        help="Number of PyTorch threads to use, default is %d." % threads_default)
    parser.add_argument("-o", "--output_dir", type=str, required=False, default="./output/",
                        help="Path to the output directory.")
    parser.add_argument("-p", "--output_prefix", type=str, required=False, default="HELEN_prediction",
                        help="Prefix for the output file. Default is: HELEN_prediction")
    parser.add_argument("-g", "--gpu_mode", default=False, action="store_true",
                        help="If set then the MI355X HIP path is used; without it the host path runs (callers x threads, like the reference).")
    parser.add_argument("-d_ids", "--device_ids", type=str, required=False, default=None,
                        help="List of gpu device ids to use for inference, e.g. 0,1,2. Default: all.")
    parser.add_argument("-c", "--callers", type=int, required=False, default=8,
                        help="Total number of callers to spawn if doing CPU inference (ignored in gpu mode).")
    # the one option the reference does not have: the arithmetic of the gate matmuls on the MI355X
    parser.add_argument("--precision", type=str, required=False, default=None, choices=["fp32", "bf16", "fp32x3"],
                        help="[MI355X] gate-matmul arithmetic with -g: fp32 (default: the reference's arithmetic on the fp32 matrix\n"
                             "cores), fp32x3 (fp32-class results from three-term bf16 splits on the bf16 matrix cores, 1.5x),\n"
                             "bf16 (bf16 operands, fp32 accumulate and state, 6x; labels differ from fp32 at ~1e-5 of the\n"
                             "positions of a trained network).  Default: $HELEN_PRECISION, else fp32.")
    return parser


def add_stitch_arguments(parser):
    parser.add_argument("-i", "--input_dir", type=str, required=True,
                        help="[REQUIRED] Path to a directory containing prediction files call consensus.")
    parser.add_argument("-o", "--output_dir", type=str, required=True,
                        help="[REQUIRED] Path to the output directory.")
    parser.add_argument("-t", "--threads", type=int, required=True, help="[REQUIRED] Number of threads.")
    parser.add_argument("-p", "--output_prefix", type=str, required=False, default="HELEN_consensus",
                        help="Prefix for the output file. Default is: HELEN_consensus")
    return parser


def add_test_arguments(parser):
    """`helen_train test` (helen/helen_train.py:87-136)."""
    parser.add_argument("--test_image_dir", type=str, required=True,
                        help="Path to directory containing labeled images for testing the models.")
    parser.add_argument("--batch_size", type=int, required=False, default=100, help="Batch size, default is 100.")
    parser.add_argument("--model_path", type=str, required=False, default="./model", help="Path of the model to load")
    parser.add_argument("--gpu_mode", action="store_true", help="Run on the MI355X HIP path (without it: the host path).")
    parser.add_argument("--print_details", action="store_true", help="Accepted for compatibility.")
    parser.add_argument("--output_dir", type=str, required=False, default="./debug_output", help="Output directory.")
    parser.add_argument("--num_workers", type=int, required=False, default=40, help="Accepted for compatibility.")
    return parser


def build_parser():
    parser = argparse.ArgumentParser(
        prog="helen", formatter_class=argparse.RawTextHelpFormatter,
        description="HELEN inference path on MI355X (gfx950): MarginPolish images -> prediction HDF5.")
    parser.add_argument("--version", default=False, action="store_true", help="Show version.")   # helen.py:261-266
    sub = parser.add_subparsers(dest="sub_command")
    add_polish_arguments(sub.add_parser("polish", help="call_consensus, then stitch"), 1)
    add_polish_arguments(sub.add_parser("call_consensus", help="generate the prediction HDF5 files"), 16)
    add_stitch_arguments(sub.add_parser("stitch", help="prediction HDF5 files -> polished FASTA"))
    add_test_arguments(sub.add_parser("test", help="evaluate a model on labeled images (helen_train test)"))
    chk = sub.add_parser("check_images", help="vet the HDF5 schema of a MarginPolish image directory")
    chk.add_argument("-i", "--image_dir", type=str, required=True, help="directory of MarginPolish .h5 images")
    chk.add_argument("--images-per-file", type=int, default=8, help="images sampled per file")
    chk.add_argument("--strict", action="store_true",
                     help="inspect EVERY image and read every image through the product reader; exit 0 = ready, "
                          "1 = schema problems, 2 = the reader refuses something")
    chk.add_argument("--json", type=str, default=None, metavar="PATH", help="write the full report as JSON ('-' = stdout)")
    sub.add_parser("version", help="show the version")
    sub.add_parser("torch_stat", help="show torch / device configuration")
    return parser


def main(argv=None):
    parser = build_parser()
    flags, unparsed = parser.parse_known_args(argv)
    if getattr(flags, "precision", None):
        import os
        os.environ["HELEN_PRECISION"] = flags.precision       # read where a rank builds its model (helen_amd/transducer.py)
    if flags.sub_command == "polish":
        from .call_consensus import polish_genome
        polish_genome(flags.image_dir, flags.model_path, flags.batch_size, flags.num_workers,
                      flags.threads, flags.output_dir, flags.output_prefix, flags.gpu_mode,
                      flags.device_ids, flags.callers)
    elif flags.sub_command == "call_consensus":
        from .call_consensus import call_consensus
        call_consensus(flags.image_dir, flags.model_path, flags.batch_size, flags.num_workers,
                       flags.threads, flags.output_dir, flags.output_prefix, flags.gpu_mode,
                       flags.device_ids, flags.callers)
    elif flags.sub_command == "stitch":
        from .stitch import perform_stitch
        perform_stitch(flags.input_dir, flags.output_dir, flags.output_prefix, flags.threads)
    elif flags.sub_command == "test":
        from .evaluate import test_interface
        sys.stderr.write("INFO: TEST MODULE SELECTED\n")
        test_interface(flags.test_image_dir, flags.batch_size, flags.gpu_mode, flags.num_workers,
                       flags.model_path, flags.output_dir, flags.print_details)
    elif flags.sub_command == "check_images":
        from .check_images import main as check_main
        return check_main(flags.image_dir, flags.images_per_file, flags.strict, flags.json)
    elif flags.sub_command == "version" or (flags.sub_command is None and flags.version):
        print("HELEN-MI355X VERSION: " + __version__)
    elif flags.sub_command == "torch_stat":
        import torch
        print("TORCH VERSION: " + torch.__version__)
        print("GPU AVAILABLE: " + str(torch.cuda.is_available()))
        if torch.cuda.is_available():
            for i in range(torch.cuda.device_count()):
                print("DEVICE %d: %s" % (i, torch.cuda.get_device_properties(i).name))
    else:
        sys.stderr.write("ERROR: NO SUBCOMMAND PROVIDED. PLEASE USE --help TO SEE THE OPTIONS.\n")
        parser.print_help(sys.stderr)
        return 1
    import os
    if os.environ.get("HELEN_ASSERT_NO_TORCH", "") == "1" and "torch" in sys.modules:
        # (tests: `polish` / `call_consensus` / `stitch` are meant to run without torch, helen_amd/native_engine.py)
        sys.stderr.write("ERROR: torch WAS IMPORTED DURING THIS RUN (HELEN_ASSERT_NO_TORCH=1).\n")
        return 3
    return 0


def leave(code):
    """End a COMMAND that has done its work: flush, run the exit handlers, and leave without the interpreter's tear-down
    (a `polish` of a chromosome holds hundreds of megabytes of sequences and a table of 100,000 joins whose
    deallocation, with the page-locked slots still on their way back to the runtime, is 0.2 s of a 5 s run).  Every file
    of the run is closed by then and every child process joined; a failure never comes here (exceptions and sys.exit
    take the ordinary way out)."""
    import atexit
    import os
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    finally:
        atexit._run_exitfuncs()
        os._exit(int(code or 0))


def entry():
    """The `helen` command (pyproject.toml, bin/helen)."""
    leave(main())


def train_entry():
    """The `helen_train` command."""
    leave(train_main())


def build_train_parser():
    """`helen_train` (helen/helen_train.py:196-223): of its sub-commands the evaluation one, `test`, is on this build's
    path; `train` is accepted by the parser so that the refusal can say why."""
    parser = argparse.ArgumentParser(
        prog="helen_train", formatter_class=argparse.RawTextHelpFormatter,
        description="Evaluation of a HELEN model on labeled MarginPolish images on MI355X (gfx950).")
    parser.add_argument("--version", default=False, action="store_true", help="Show version.")
    sub = parser.add_subparsers(dest="sub_command")
    add_test_arguments(sub.add_parser("test", help="Test a model. Requires a set of labeled images"))
    sub.add_parser("train", help="(not part of this build: training is outside the inference path)", add_help=False)
    sub.add_parser("torch_stat", help="See PyTorch configuration.")
    sub.add_parser("version", help="Show program version.")
    return parser


def train_main(argv=None):
    """Console script `helen_train` (setup.py:155 of the reference)."""
    parser = build_train_parser()
    flags, unparsed = parser.parse_known_args(argv)
    if flags.sub_command == "test":
        args = list(sys.argv[1:] if argv is None else argv)
        args.remove("test")
        return main(["test"] + args)
    if flags.sub_command == "train":
        sys.stderr.write("ERROR: `helen_train train` IS NOT PART OF THIS BUILD (the MI355X inference path of `helen polish`; "
                         "`helen_train test` evaluates a trained model).\n")
        return 1
    if flags.sub_command == "torch_stat":
        return main(["torch_stat"])
    if flags.sub_command == "version" or flags.version:
        return main(["version"])
    sys.stderr.write("ERROR: NO SUBCOMMAND PROVIDED. PLEASE USE --help TO SEE THE OPTIONS.\n")
    parser.print_help(sys.stderr)
    return 1


if __name__ == "__main__":
    entry()
