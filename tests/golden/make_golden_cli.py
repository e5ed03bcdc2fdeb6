"""TEST INFRASTRUCTURE -- generates tests/golden/cli_ref.json from the reference's own command line definition.

Runs only in the build container (needs /root/reference):

    python tests/golden/make_golden_cli.py

The reference's `helen/helen.py` is imported and its `add_polish_arguments`, `add_call_consensus_arguments` and
`add_stitch_arguments` (helen.py:12-222) and `helen/helen_train.py`'s `add_test_arguments` (helen_train.py:87-136) are
applied to fresh argparse parsers; the fixture lists, per sub-command and in
order, every option's flags, destination, default, type, `required`, `nargs` and `const` (help texts are not compared).
Importing helen.py pulls in the whole package: beside tests/golden/reference_env.py's environment, `onnx`, `onnxruntime`
(the reference's CPU path), `wget` (its model downloader), `torchnet` and `matplotlib` (its training / evaluation
reports) are registered as EMPTY placeholder modules -- nothing of them
is called when option tables are built.
"""
import argparse
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cli_ref.json")


def option_table(parser):
    return [{"flags": a.option_strings, "dest": a.dest, "default": a.default, "required": a.required,
             "type": getattr(a.type, "__name__", None), "nargs": a.nargs, "const": a.const}
            for a in parser._actions if a.dest != "help"]


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference")
    install()
    for name in ("onnx", "onnxruntime", "wget", "torchnet", "torchnet.meter", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchnet"].meter = sys.modules["torchnet.meter"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["matplotlib"].use = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    from helen import helen as reference_cli                                   # the reference's own module
    out = {}
    for sub, add in (("polish", reference_cli.add_polish_arguments),
                     ("call_consensus", reference_cli.add_call_consensus_arguments),
                     ("stitch", reference_cli.add_stitch_arguments)):
        parser = argparse.ArgumentParser()
        add(parser)
        out[sub] = option_table(parser)
    from helen import helen_train as reference_train_cli                       # helen_train.py:87-136: `helen_train test`
    parser = argparse.ArgumentParser()
    reference_train_cli.add_test_arguments(parser)
    out["helen_train test"] = option_table(parser)
    with open(OUT, "w") as f:
        json.dump({"made_by": "tests/golden/make_golden_cli.py (reference helen/helen.py imported)", "options": out}, f,
                  indent=1, sort_keys=True)
    print("wrote %s: %s" % (OUT, {k: len(v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
