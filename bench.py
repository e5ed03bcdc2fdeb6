#!/usr/bin/env python3
"""bench.py -- pileup windows / second of the `helen polish` inference path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`, one rank per GPU.  N > 1 either arrives
under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, started plainly,
re-executes itself under `python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1 -- and
refuses when fewer than N GPUs are visible.  A "step" is one pass of the hot path -- one helen_polish_batch call -- over
`--coalesce` (16) loader batches of 256 synthetic pileup windows (1000 positions x 90 features,
uint8, already resident in HBM): uint8->f32, 19 overlapping chunks of the 2-layer bidirectional GRU
with carried hidden state, heads, softmax-accumulate and argmax labels (reference:
models/predict_gpu.py:97-159).  Windows are independent and hidden is zeroed per window, so handing
the device 16 loader batches at once gives the same labels as 16 separate calls
(tests/test_gpu_parity.py::test_batch_split_invariance); 4096 windows = 256 tiles is what fills
256 CUs x 2 workgroups.  `value` is windows per second over all ranks.

Prints ONE JSON line (rank 0) with the whole-job windows/s (`value`: inputs resident in HBM when the
timed region starts), the MFMA-roofline figures of the dominant kernel (the GRU recurrence, timed
with recycled HIP events on the launch stream inside the timed region), `host_path` -- the same
windows from page-locked HOST memory to labels in HOST memory through helen_polish_host (PCIe
included, SURVEY.md 8d's "images in host memory -> labels in host memory"; reported beside `value`,
never as it) -- per-rank rates, and a CPU baseline (the repo's oracle, a port of the reference
algorithm, on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_WINDOW = 1772441600.0        # 932,864 FLOP/step x 100 steps x 19 chunks
# One recurrence launch, both directions: h.W_hh^T (100 steps x 2 x 384 x 128 MACs); the decoder launches
# also carry the heads' product (8,192 FLOP per timestep).  Averaged over the encoder and
# decoder launches, which the timing below also averages.
GRU_FLOP_PER_WINDOW_LAUNCH = 100 * 2 * 2.0 * 384 * 128 + 100 * 8192 / 2.0
# SURVEY.md 8d's matmul FLOPs per GRU timestep, both directions, by where a call spends them
ENC_PROJ_FLOP_STEP = 2 * 2.0 * 384 * 90       # x_t . W_ih^T, encoder
ENC_REC_FLOP_STEP = 2 * 2.0 * 384 * 128       # h . W_hh^T, encoder
DEC_PROJ_FLOP_STEP = 2 * 2.0 * 384 * 256      # y1_t . W_ih^T, decoder
DEC_REC_FLOP_STEP = 2 * 2.0 * 384 * 128       # h . W_hh^T, decoder
HEADS_FLOP_STEP = 2.0 * 256 * 16              # [h_fwd | h_bwd] . [W_base ; W_rle]^T
# one fused bf16 layer launch (projection + recurrence; the decoder's also the heads), averaged over the two layers
BF16_LAYER_FLOP_PER_WINDOW_LAUNCH = 100 * (ENC_PROJ_FLOP_STEP + ENC_REC_FLOP_STEP + DEC_PROJ_FLOP_STEP + DEC_REC_FLOP_STEP
                                           + HEADS_FLOP_STEP) / 2.0
FP32_MFMA_PEAK = 157.3e12             # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK = 2.5e15               # MI355X_MICROARCH.md: bf16 MFMA, dense


NOMINAL_SCLK_MHZ = 2400.0             # MI355X_MICROARCH.md: the clock the MFMA peaks are quoted at


def _amdgpu_hwmon():
    """[(PCI address, hwmon directory)] of every amdgpu card that publishes its shader clock and socket power."""
    import glob
    out = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(h + "/freq1_input") and os.path.exists(h + "/power1_input"):
            out.append((os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(h)))), h))
    return out


def sustained_clock(run, dev, seconds=1.5):
    """Shader clock and socket power while `run(4)` repeats for `seconds`, read from the amdgpu hwmon files by a side thread
    every 10 ms (OUTSIDE every timed region).  The bf16-pipe modes sit at the part's power cap and run at a lower clock than
    the 2.4 GHz their peaks are quoted at (profiles/r06_clock_power.txt): the roofline fractions stay against the nominal
    peaks, this says what the device sustained under this very load.  Never raises: a box without the files says so."""
    import threading
    try:
        import torch
        cards = _amdgpu_hwmon()
        if not cards:
            return {"error": "no amdgpu hwmon freq1_input / power1_input under /sys/class/drm"}
        props = torch.cuda.get_device_properties(dev)
        want = None
        if hasattr(props, "pci_bus_id"):
            want = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
        pick = [c for c in cards if c[0] == want] or cards
        seen = {c[0]: [] for c in pick}
        stop = threading.Event()

        def watch():
            while not stop.is_set():
                for pci, h in pick:
                    try:
                        with open(h + "/freq1_input") as f:
                            mhz = int(f.read()) / 1e6
                        with open(h + "/power1_input") as f:
                            watt = int(f.read()) / 1e6
                        seen[pci].append((time.perf_counter(), mhz, watt))
                    except (OSError, ValueError):
                        pass
                time.sleep(0.01)

        th = threading.Thread(target=watch, daemon=True)
        run(4)
        torch.cuda.synchronize(dev)
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            run(4)
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        stop.set()
        th.join(timeout=2.0)
        # the card under this load = the one asked for by PCI address, else the one drawing the most power meanwhile
        best, rows = None, []
        for pci, vals in seen.items():
            vals = [v for v in vals if t0 + 0.3 <= v[0] <= t1]
            if vals and (best is None or sum(v[2] for v in vals) / len(vals) > sum(v[2] for v in rows) / len(rows)):
                best, rows = pci, vals
        if not rows:
            return {"error": "no readable samples"}
        mhz = sorted(v[1] for v in rows)[len(rows) // 2]
        watt = sorted(v[2] for v in rows)[len(rows) // 2]
        cap = None
        try:
            with open(dict(pick)[best] + "/power1_cap") as f:
                cap = int(f.read()) / 1e6
        except (OSError, ValueError):
            pass
        return {"sclk_mhz": round(mhz, 0), "socket_power_w": round(watt, 0), "power_cap_w": cap, "nominal_mhz": NOMINAL_SCLK_MHZ,
                "samples": len(rows), "seconds": round(t1 - t0, 2), "card": best,
                "matched_by": "PCI address" if want is not None and best == want else "highest power draw",
                "source": "amdgpu hwmon freq1_input (sclk) / power1_input (PPT), medians over back-to-back calls outside the timed region"}
    except Exception as e:      # noqa: BLE001 -- a side report: it must not be able to take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def with_clock(roofline, clock):
    """`roofline` + the clock report and the fractions read against the sustained clock instead of the nominal one."""
    roofline["clock"] = clock
    mhz = clock.get("sclk_mhz") if isinstance(clock, dict) else None
    if mhz and roofline.get("bound") == "mfma":
        roofline["frac_at_sustained_clock"] = round(roofline["frac"] * NOMINAL_SCLK_MHZ / mhz, 4)
        if roofline.get("path_frac") is not None:
            roofline["path_frac_at_sustained_clock"] = round(roofline["path_frac"] * NOMINAL_SCLK_MHZ / mhz, 4)
    return roofline


def pmc_traffic(windows_per_launch, precision="fp32"):
    """HBM bytes per recurrence launch from the committed rocprofv3 PMC summary of this same command
    (profiles/*_pmc_summary.json, FETCH_SIZE/WRITE_SIZE passes; see scripts/pmc_summary.py), scaled to this run's
    windows per launch.  fp32: the newest summary without a mode tag, kernels gru_kernel / gru_pair_kernel; bf16: the
    newest `*_bf16_pmc_summary.json`, the fused layer kernels.  None if no summary is committed."""
    import glob
    tag = {"fp32": None, "bf16": "bf16", "fp32x3": "fp32x3"}[precision]
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))
                   if (tag in os.path.basename(f) if tag else not any(t in os.path.basename(f) for t in ("bf16", "fp32x3"))))
    if not files:
        return None, None
    names = {"bf16": ("helen::gru_fused_bf16",), "fp32x3": ("helen::gru_x3",),
             "fp32": ("helen::gru_kernel", "helen::gru_pair_kernel")}[precision]
    try:
        # encoder and decoder launches of the recurrence, launch-weighted; the two-tile kernels where both were profiled
        ks = {k: v for k, v in json.load(open(files[-1]))["kernels"].items() if k.startswith(names)}
        if any("pair" in k or "_il_" in k for k in ks):
            ks = {k: v for k, v in ks.items() if "pair" in k or "_il_" in k}
        n = sum(k["launches_profiled"] for k in ks.values())
        per_launch = sum(k["hbm_bytes_per_launch"] * k["launches_profiled"] for k in ks.values()) / n
        base = 8192.0 if precision == "bf16" else 4096.0          # windows per launch of the profiled command
        return int(per_launch * windows_per_launch / base), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_call_bytes_per_window(precision="fp32"):
    """HBM bytes per window of a WHOLE device call, every kernel of the call summed, from the same committed counter
    summary pmc_traffic reads (launches per call: pack and the encoder projection once, the rest per chunk)."""
    import glob
    tag = {"fp32": None, "bf16": "bf16", "fp32x3": "fp32x3"}[precision]
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))
                   if (tag in os.path.basename(f) if tag else not any(t in os.path.basename(f) for t in ("bf16", "fp32x3"))))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        per_call = doc.get("call_bytes_per_window")
        return int(per_call) if per_call else None
    except Exception:
        return None


def kernel_table(all_stats, calls, call_windows, precision, ms_per_step):
    """`roofline.kernels`: every kernel class of a device call with its share of a step and its own roofline fraction,
    from HIP events around EVERY launch of `calls` extra (untimed-for-the-headline) calls on the launch stream.
    Algorithmic FLOPs per window and launch are SURVEY.md 8d's, split by where the call spends them; `pipe` is the matrix
    pipe the class's kernels issue to in this precision."""
    enc_all = 1000 * ENC_PROJ_FLOP_STEP                 # the encoder projection is computed ONCE for all 1000 positions
    if precision == "bf16":
        flop = {"gru_enc": 100 * (ENC_PROJ_FLOP_STEP + ENC_REC_FLOP_STEP),
                "gru_dec": 100 * (DEC_PROJ_FLOP_STEP + DEC_REC_FLOP_STEP + HEADS_FLOP_STEP)}
        peak, pipe = BF16_MFMA_PEAK, "bf16 MFMA (v_mfma_f32_16x16x32_bf16 / 32x32x16), fp32 accumulate"
    else:
        flop = {"gemm_enc": enc_all, "gru_enc": 100 * ENC_REC_FLOP_STEP, "gemm_dec": 100 * DEC_PROJ_FLOP_STEP,
                "gru_dec": 100 * (DEC_REC_FLOP_STEP + HEADS_FLOP_STEP)}
        if precision == "fp32x3":
            peak, pipe = X3_MFMA_PEAK, "bf16 MFMA, 6 products of 3-term splits per fp32 product (peak = bf16 / 6)"
        else:
            peak, pipe = FP32_MFMA_PEAK, "fp32 MFMA (v_mfma_f32_16x16x4_f32 / 4x4x1)"
    # HBM-bound classes (fp32 / fp32x3): pack = uint8 image in, bf16 A fragments out (K padded to 96); the encoder projection
    # runs on the bf16 pipe with exact products (counts are one bf16 term, W_ih three: 3 MFMAs per 32 k, fp32 accumulation)
    # and is bound by its fp32 output stream: 192 KB of fragments in, 2 x 384 x 1000 x 4 B of gi out per window
    hbm_bytes = {"pack": 90000.0 + 96 * 2 * 1000.0, "gemm_enc": 96 * 2 * 1000.0 + 2 * 384 * 1000 * 4.0}
    rows = []
    for name, (ms, n) in all_stats.items():
        if n == 0:
            continue
        avg = ms / n
        per_step = ms / calls
        row = {"class": name, "launches_per_call": round(n / float(calls), 2), "avg_launch_ms": round(avg, 4),
               "ms_per_step": round(per_step, 3), "share_of_step": round(per_step / ms_per_step, 4)}
        if name in hbm_bytes and precision != "bf16":
            gb = hbm_bytes[name] * call_windows / (avg * 1e-3) / 1e9
            row.update({"bound": "hbm", "achieved_GBps": round(gb, 1), "frac": round(gb / 8000.0, 4)})
            if name in flop:
                tf = flop[name] * call_windows / (avg * 1e-3) / 1e12
                row.update({"pipe": "bf16 MFMA, exact products: pileup counts (one bf16 term) x W_ih in three bf16 terms, fp32 accumulate",
                            "flop_per_window_launch": flop[name], "achieved_TFLOPs": round(tf, 1),
                            "frac_of_fp32_mfma_peak": round(tf * 1e12 / FP32_MFMA_PEAK, 4)})
        elif name in flop:
            tf = flop[name] * call_windows / (avg * 1e-3) / 1e12
            row.update({"bound": "mfma", "pipe": pipe, "flop_per_window_launch": flop[name], "achieved_TFLOPs": round(tf, 1),
                        "frac": round(tf * 1e12 / peak, 4)})
        else:
            row.update({"bound": "latency" if name == "heads" else "hbm", "frac": None})
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def usable_cpus():
    from helen_amd.host_plan import usable_cpus as f
    return f()


def cpu_baseline(batch, seconds_target=12.0):
    """Time the CPU oracle (port of the reference path) on this box's host cores, bounded."""
    import oracle
    from helen_amd.weights import make_images, make_weights
    threads = min(oracle.max_threads(), usable_cpus())
    oracle.set_threads(threads)
    w = make_weights(input_scale=1.0 / 64.0)
    probe = make_images(threads * 8, seed=1)   # one 8-window block per thread
    t0 = time.time()
    oracle.polish_batch(w, probe)
    dt = time.time() - t0
    rate = probe.shape[0] / dt
    n = int(min(max(rate * seconds_target, probe.shape[0]), 16 * batch))
    n = max(8, (n // 8) * 8)
    img = make_images(n, seed=2)
    t0 = time.time()
    oracle.polish_batch(w, img)
    dt = time.time() - t0
    return {"value": round(n / dt, 2), "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": "%d uniform-random windows through oracle/helen_oracle.c (fp32, OpenMP over "
                      "8-window blocks, %d threads = usable CPUs under the cgroup quota), %.1f s" % (n, threads, dt)}


def torch_eager_baseline(batch, seconds_target=10.0):
    """What the reference's arithmetic library does on this box's host cores: torch.nn.GRU + Linear of the reference's
    architecture (TransducerModel.py:43-58) driven by the reference's loop (predict_gpu.py:99-159: 19 chunks of 100
    positions at stride 50, hidden carried, softmax, zero-pad-add, argmax) at the loader batch, eager, fp32, on the CPU with
    every usable thread -- a RESTATEMENT written here (the reference's modules cannot travel to this box and its CLI's
    CPU path is an ONNX Runtime session, which this image does not have); checked against the oracle's labels on the
    sample's first windows.  The closest this box can come to "the reference CPU path on the same cores"."""
    import numpy as np
    import torch

    import oracle
    from helen_amd.weights import make_images, make_weights
    threads = usable_cpus()
    torch.set_num_threads(threads)
    w = make_weights(input_scale=1.0 / 64.0)
    enc = torch.nn.GRU(90, 128, num_layers=1, bidirectional=True, batch_first=True)
    dec = torch.nn.GRU(256, 128, num_layers=1, bidirectional=True, batch_first=True)
    base, rle = torch.nn.Linear(256, 5), torch.nn.Linear(256, 11)
    with torch.no_grad():
        for mod, prefix in ((enc, "gru_encoder."), (dec, "gru_decoder."), (base, "dense1_base."), (rle, "dense2_rle.")):
            for name, par in mod.named_parameters():
                par.copy_(torch.from_numpy(w[prefix + name]))

    def polish(images):                       # uint8 [n, 1000, 90] -> labels, batch by batch
        out_b, out_r = [], []
        with torch.no_grad():
            for lo in range(0, images.shape[0], batch):
                x = torch.from_numpy(images[lo:lo + batch]).float()
                hidden = torch.zeros(2, x.shape[0], 128)
                acc_b = torch.zeros(x.shape[0], 1000, 5)
                acc_r = torch.zeros(x.shape[0], 1000, 11)
                for i in range(0, 901, 50):
                    y1, h1 = enc(x[:, i:i + 100], hidden)
                    y2, hidden = dec(y1, h1)
                    acc_b[:, i:i + 100] += torch.softmax(base(y2), dim=2)
                    acc_r[:, i:i + 100] += torch.softmax(rle(y2), dim=2)
                out_b.append(acc_b.argmax(dim=2).numpy().astype(np.uint8))
                out_r.append(acc_r.argmax(dim=2).numpy().astype(np.uint8))
        return np.concatenate(out_b), np.concatenate(out_r)

    probe = make_images(min(batch, 64), seed=5)
    t0 = time.time()
    pb, pr = polish(probe)
    rate = probe.shape[0] / (time.time() - t0)
    ref = oracle.polish_batch(w, probe[:8])
    differ = int((pb[:8] != ref["bases"]).sum() + (pr[:8] != ref["rles"]).sum())
    n = int(min(max(rate * seconds_target, 64), 4 * batch))
    n = n // batch * batch if n >= batch else n // 8 * 8      # whole loader batches, or one short one on a slow host
    img = make_images(n, seed=6)
    t0 = time.time()
    polish(img)
    dt = time.time() - t0
    return {"value": round(n / dt, 2), "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": "%d uniform-random windows at batch %d through torch %s nn.GRU / Linear eager on the CPU (a restatement of "
                      "TransducerModel.py:43-79 + predict_gpu.py:99-159), %d threads, %.1f s" % (n, batch, torch.__version__, threads, dt),
            "labels_differing_from_the_oracle_on_8_windows": differ}


def host_mode(seconds_target=8.0):
    """The product's OWN path for runs without --gpu_mode (helen_amd/csrc/cpu_path.cpp through helen_amd/cpu_engine.py: not
    the oracle), on a bounded sample with every usable thread: what `helen polish` without -g does per caller."""
    from helen_amd.cpu_engine import CpuEngine
    from helen_amd.weights import make_images, make_weights
    threads = usable_cpus()
    eng = CpuEngine(make_weights(input_scale=1.0 / 64.0), threads=threads)
    probe = make_images(threads * 16, seed=3)          # one 16-window block per thread
    t0 = time.time()
    eng.polish_host(probe)
    rate = probe.shape[0] / (time.time() - t0)
    n = max(16, int(min(max(rate * seconds_target, probe.shape[0]), 8192)) // 16 * 16)
    img = make_images(n, seed=4)
    t0 = time.time()
    eng.polish_host(img)
    dt = time.time() - t0
    eng.close()
    return {"value": round(n / dt, 2), "unit": "windows/s", "threads": threads,
            "what": "libhelen_cpu.so (the product's host path: fp32, OpenMP over 16-window blocks), %d uniform-random windows, "
                    "%.1f s; labels pinned to the same goldens as the GPU path (tests/test_cpu_path.py)" % (n, dt)}


def call_size_report(images, dev):
    """Throughput of ONE fp32 device call of fewer windows than the headline's 4096 (inputs resident in HBM; 1 warm-up call,
    then 6 timed ones): which kernels a call takes depends on how many tiles it has (quarter / half tiles on
    v_mfma_f32_4x4x1_16b_f32 up to 32 / 64 tiles, one 8-wave workgroup per (tile, direction) up to 128, two tile groups on
    two streams up to 239, tile pairs above: DESIGN.md 4 and 6).  All choices give the same bits (tests/test_gpu_scale.py)."""
    import torch
    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_weights
    out = {"unit": "windows/s", "what": call_size_report.__doc__.split(":")[0].replace("\n    ", " ")}
    for n in (256, 512, 1024, 2048, 3072):
        if images.shape[0] < n:
            continue
        e = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=dev.index, max_windows=n)
        try:
            e.polish(images[:n])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(6):
                e.polish(images[:n])
            torch.cuda.synchronize(dev)
            out[str(n)] = round(6 * n / (time.perf_counter() - t0), 1)
        finally:
            e.close()
    return out


def margin_report(images, dev, precision):
    """Which labels can move: the top-1 / top-2 margin histogram of the fp32 path's accumulated softmax over up to
    4096 windows of the shard, for the bench's random-init network (heads x8) and for the `peaked` stand-in of a
    trained model (heads x64, helen_amd.weights.make_peaked_weights) -- and, for a reduced-precision mode, the share
    of labels it has in common with the fp32 path on each.  `below_2e-06` is the fp32 tie rate: where two correct
    fp32 evaluations may call different labels (tests/test_gpu_scale.py: 3.7e-7 of the labels do)."""
    import numpy as np

    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_peaked_weights, make_weights, margin_histogram
    n = min(images.shape[0], 4096)
    out = {"windows": n, "what": "fractions of positions with top1-top2 margin of the accumulated softmax below each "
                                 "edge (fp32 path), per head"}
    cases = [("random_init_heads_x8", make_weights(input_scale=1.0 / 64.0), images[:n], None),
             ("peaked_heads_x64", make_peaked_weights(), images[:n], None)]
    trained = os.path.join(ROOT, "tests", "golden", "trained_synth.npz")
    if os.path.exists(trained):
        # the reference model TRAINED on a synthetic polishing task (tests/golden/make_trained_synth.py), on 512 fresh
        # windows of that task: the only network here whose outputs are confident because it learnt something
        import torch

        from helen_amd.synthetic import make_pileup_task
        z = np.load(trained)
        timg, tlb, tlr = make_pileup_task(512, seed=int(z["_task_seed"]) + 9000)
        cases.append(("trained_on_synthetic_task", {k: z[k] for k in z.files if not k.startswith("_")},
                      torch.from_numpy(timg).to(dev), (tlb, tlr)))
    for name, w, imgs, truth in cases:
        ref = HelenEngine(w, device=dev.index, max_windows=n, precision="fp32")
        b0, r0, ab, ar = ref.polish(imgs, want_acc=True)
        entry = {"base": margin_histogram(ab.cpu().numpy()), "rle": margin_histogram(ar.cpu().numpy())}
        entry["median_top1_of_2"] = {"base": round(float(np.median(ab.max(-1).values.cpu().numpy())), 4),
                                     "rle": round(float(np.median(ar.max(-1).values.cpu().numpy())), 4)}
        ref.close()
        entry["windows"] = int(imgs.shape[0])
        if truth is not None:
            entry["accuracy_vs_truth"] = {"base": round(float((b0.cpu().numpy() == truth[0]).mean()), 5),
                                          "rle": round(float((r0.cpu().numpy() == truth[1]).mean()), 5)}
        if precision != "fp32":
            alt = HelenEngine(w, device=dev.index, max_windows=n, precision=precision)
            b1, r1 = alt.polish(imgs)
            entry["label_identity_vs_fp32"] = {"base": round(float((b0 == b1).float().mean().item()), 6),
                                               "rle": round(float((r0 == r1).float().mean().item()), 6)}
            if truth is not None:
                entry["accuracy_vs_truth_" + precision] = {"base": round(float((b1.cpu().numpy() == truth[0]).mean()), 5),
                                                            "rle": round(float((r1.cpu().numpy() == truth[1]).mean()), 5)}
            alt.close()
        out[name] = entry
    return out


def precision_check(eng, precision, images, dev):
    """BASELINE.json configs[3] asks for the reduced-precision mode's own check: argmax parity and logits tolerance
    against the fp32 path on the same inputs.  Labels: one device call of each engine over the first windows of the
    shard.  Logits: the reference's 19-chunk operator loop (predict_gpu.py:114-129, hidden carried) on 512 of them,
    both engines, max |logit difference| over all chunks.  ALL WEIGHTS ARE SYNTHETIC (no trained model exists
    offline): `label_identity` is on the bench's random-init network, whose margins are thin; `margins` repeats it on
    the peaked stand-in of a trained model."""
    import torch

    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_weights
    n = min(images.shape[0], 4096)
    ref = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=dev.index, max_windows=n, precision="fp32")
    b0, r0 = ref.polish(images[:n])
    b1, r1 = eng.polish(images[:n])
    same = float((b0 == b1).float().mean().item() + (r0 == r1).float().mean().item()) / 2
    m = min(n, 512)
    x = images[:m].float()
    h0 = torch.zeros((m, 2, 128), device=dev)
    h1 = h0.clone()
    worst, biggest = 0.0, 0.0
    for c in range(19):
        xc = x[:, 50 * c:50 * c + 100].contiguous()
        ob0, or0, h0 = ref.chunk_forward(xc, h0)
        ob1, or1, h1 = eng.chunk_forward(xc, h1)
        worst = max(worst, float((ob0 - ob1).abs().max().item()), float((or0 - or1).abs().max().item()))
        biggest = max(biggest, float(ob0.abs().max().item()), float(or0.abs().max().item()))
    ref.close()
    return {"against": "the fp32 path of this library on the same windows", "weights": "synthetic random-init (heads x8); "
            "see `margins` for the peaked stand-in of a trained model", "windows_labels": n,
            "label_identity": round(same, 6), "windows_logits": m, "max_abs_logit_diff": float("%.3g" % worst),
            "max_abs_logit": round(biggest, 3), "precision": precision}


X3_MFMA_PEAK = BF16_MFMA_PEAK / 6.0    # fp32x3: six bf16 MFMAs per fp32 product group -> 416.7 TFLOP/s of fp32-class work


def trained_identity(dev, precision):
    """Labels of `precision` against the fp32 path on the reference model TRAINED on a synthetic polishing task
    (tests/golden/trained_synth.npz, made by tests/golden/make_trained_synth.py): 512 fresh windows of that task."""
    import numpy as np
    import torch

    from helen_amd.engine import HelenEngine
    from helen_amd.synthetic import make_pileup_task
    trained = os.path.join(ROOT, "tests", "golden", "trained_synth.npz")
    if not os.path.exists(trained):
        return None
    z = np.load(trained)
    w = {k: z[k] for k in z.files if not k.startswith("_")}
    timg, tlb, tlr = make_pileup_task(512, seed=int(z["_task_seed"]) + 9000)
    imgs = torch.from_numpy(timg).to(dev)
    out = {"windows": 512}
    labels = {}
    for prec in ("fp32", precision):
        e = HelenEngine(w, device=dev.index, max_windows=512, precision=prec)
        b, r = e.polish(imgs)
        labels[prec] = (b.cpu().numpy(), r.cpu().numpy())
        e.close()
        out["accuracy_vs_truth_" + prec] = {"base": round(float((labels[prec][0] == tlb).mean()), 5),
                                            "rle": round(float((labels[prec][1] == tlr).mean()), 5)}
    out["label_identity_vs_fp32"] = {"base": round(float((labels["fp32"][0] == labels[precision][0]).mean()), 6),
                                     "rle": round(float((labels["fp32"][1] == labels[precision][1]).mean()), 6)}
    return out


def mode_report(precision, batch, coalesce, images, dev, fp32_labels, steps=8, warmup=2):
    """One of the library's other arithmetic modes under the same clock as the headline: `steps` device calls of
    `batch` x `coalesce` windows of the resident shard (inputs in HBM, wall clock between device syncs), the recurrence
    launches timed by HIP events inside the timed region, its labels against the fp32 path's on the same windows, the
    operator loop's logits against fp32's, and the trained-network identity.  bf16 at batch 512 is BASELINE.json
    configs[3]; fp32x3 is the opt-in fp32-class mode on the bf16 matrix cores."""
    import ctypes

    import torch

    from helen_amd import _lib
    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_weights
    lib = _lib.load()
    call_windows = batch * coalesce
    n_calls_res = images.shape[0] // call_windows
    if n_calls_res < 1:
        return {"value": None, "skipped": "the resident shard holds fewer than %d windows" % call_windows}
    eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=dev.index, max_windows=call_windows, precision=precision)
    try:
        bases = torch.empty((n_calls_res * call_windows, 1000), dtype=torch.uint8, device=dev)
        rles = torch.empty_like(bases)
        stream = torch.cuda.current_stream(dev).cuda_stream

        def run(n):
            for k in range(n):
                s = (k % n_calls_res) * call_windows
                e = s + call_windows
                _lib.check(lib.helen_polish_batch(eng._handle, images[s:e].data_ptr(), call_windows, bases[s:e].data_ptr(),
                                                  rles[s:e].data_ptr(), None, None, ctypes.c_void_p(stream)))
        eng.set_profiling(["gru_enc", "gru_dec"])
        run(max(warmup, steps))               # untimed; also creates the event pairs the timed steps recycle
        torch.cuda.synchronize(dev)
        eng.reset_kernel_stats()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        stats = eng.kernel_stats()
        eng.set_profiling(list(_lib.KERNEL_CLASSES))          # every class, launch by launch, outside the timed region
        run(1)
        torch.cuda.synchronize(dev)
        eng.reset_kernel_stats()
        run(4)
        torch.cuda.synchronize(dev)
        all_stats = eng.kernel_stats()
        eng.set_profiling([])
        clock = sustained_clock(run, dev)
        value = steps * call_windows / elapsed
        n_l = stats["gru_enc"][1] + stats["gru_dec"][1]
        avg_ms = (stats["gru_enc"][0] + stats["gru_dec"][0]) / max(n_l, 1)
        if precision == "bf16":
            peak, flop = BF16_MFMA_PEAK, BF16_LAYER_FLOP_PER_WINDOW_LAUNCH
            kernel = ("gru_fused_bf16_* (projection + recurrence per layer, bf16 MFMA, fp32 accumulate / state / gates); "
                      "peak = dense bf16 MFMA")
        else:
            peak, flop = X3_MFMA_PEAK, GRU_FLOP_PER_WINDOW_LAUNCH
            kernel = ("gru_x3_kernel (recurrence; every fp32 product group as 6 bf16 MFMAs, fp32 accumulate); peak = dense "
                      "bf16 MFMA / 6")
        achieved = flop * call_windows / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(call_windows, precision)
        k = min(steps, n_calls_res) * call_windows
        k = min(k, fp32_labels[0].shape[0])
        same_b = float((bases[:k] == fp32_labels[0][:k]).float().mean().item())
        same_r = float((rles[:k] == fp32_labels[1][:k]).float().mean().item())
        out = {"value": round(value, 1), "unit": "windows/s", "precision": precision, "batch": batch,
               "batches_per_step": coalesce, "windows_per_step": call_windows, "steps": steps, "warmup": max(warmup, steps),
               "ms_per_step": round(elapsed * 1e3 / steps, 4),
               "roofline": {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": round(peak / 1e12, 1),
                            "unit": "TFLOP/s", "frac": round(achieved * 1e12 / peak, 4), "avg_launch_ms": round(avg_ms, 4),
                            "avg_launch_ms_encoder": round(stats["gru_enc"][0] / max(stats["gru_enc"][1], 1), 4),
                            "avg_launch_ms_decoder": round(stats["gru_dec"][0] / max(stats["gru_dec"][1], 1), 4),
                            "launches": n_l, "traffic": traffic, "traffic_source": traffic_src,
                            "path_frac": round(value * FLOP_PER_WINDOW / peak, 4),
                            "kernels": kernel_table(all_stats, 4, call_windows, precision, elapsed * 1e3 / steps),
                            "path_frac_of_fp32_mfma_peak": round(value * FLOP_PER_WINDOW / FP32_MFMA_PEAK, 4)},
               "label_identity": {"base": round(same_b, 6), "rle": round(same_r, 6), "windows": k,
                                  "against": "the fp32 path's labels of the same windows (the headline run above)"}}
        with_clock(out["roofline"], clock)
        chk = precision_check(eng, precision, images, dev)
        out["max_abs_logit_diff"] = chk["max_abs_logit_diff"]
        out["max_abs_logit"] = chk["max_abs_logit"]
        out["windows_logits"] = chk["windows_logits"]
    finally:
        eng.close()
    out["trained_network"] = trained_identity(dev, precision)
    return out


E2E_DEFAULT_WINDOWS = 300000       # per rank: BASELINE.json configs[1]'s chr20-scale image set (27 GB of images in /dev/shm,
                                   # written in ~25 s; the leg shrinks to what /dev/shm holds)
E2E_FILES_PER_RANK = 16


def gather_strings(dist, world, text):
    """Every rank's `text`, over CPU tensors (gloo) whatever other backend the group has."""
    import torch
    raw = torch.tensor(list(text.encode()[:4000]), dtype=torch.uint8)
    n = torch.tensor([raw.numel()], dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    padded = torch.zeros(4000, dtype=torch.uint8)
    padded[:raw.numel()] = raw
    every = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(every, padded)
    return [bytes(e[:int(k.item())].tolist()).decode(errors="replace") for e, k in zip(every, sizes)]


def trained_weights():
    """The weights of tests/golden/trained_synth.npz: the reference's TransducerGRU trained (build container,
    tests/golden/make_trained_synth.py) on the read-vote noise model the simulated assembly is rendered with."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "trained_synth.npz"))
    return {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files if not k.startswith("_")}


# what the end-to-end leg costs in RAM besides its RAM-backed files, measured on the GPU box with scripts/dev/watch_box.sh
# (profiles/r06_scale_single_device.json, N = 2): the input generator's worker processes hold a whole file each while they
# render it -- 5.3 GB per worker at 18,750 windows per file = 2.5 x the file's image + position bytes, 85 GB over 16 workers
# -- and a bench rank keeps ~6 GB (device context, heaps, page-locked host-path buffers) while it waits for rank 0
E2E_GENERATOR_TRANSIENT = 3.0 * 114152      # bytes of worker RSS per window of the file being rendered (2.5 measured)
E2E_RANK_RESIDENT_BYTES = 8 << 30           # per bench rank (6 GB measured)
E2E_RAM_FRACTION = 0.8                      # of what the process tree may still take (MemAvailable / cgroup head-room)


def e2e_generator_processes(world, cpus=None):
    """Worker processes PER RANK of the synthetic-input generation."""
    return max(1, min(8, (usable_cpus() if cpus is None else cpus) // world))


def e2e_size(windows_per_rank, world, may_shrink, free=None, ram=None, cpus=None):
    """What the end-to-end leg will run with: (windows per rank, bytes of RAM-backed space it needs, where it goes, the
    tmpfs budget).  The default leg (no --e2e given) shrinks -- down to two device calls per rank -- until BOTH hold:
    its RAM-backed files fit the tmpfs budget (`free`: /dev/shm's free space and no more than half of the RAM the process
    tree may still take), and files + the generator workers' transient memory + the bench ranks' own resident memory
    fit 0.8 of that RAM (`ram`; None = unknown, not checked).  Round 5 sized the leg by statvfs alone and lost a box at
    N = 2; sizing it by the tmpfs budget alone leaves N = 8 on a 300 GiB cgroup at 160 GB of files + 85 GB of generator
    workers + 8 x 6 GB of ranks + the 8 call_consensus processes' slots: over the limit.  With one rank the
    leg may go to the temp directory instead of /dev/shm; otherwise it is skipped."""
    from helen_amd.host_plan import SLOT_BYTES_PER_WINDOW, ram_available_bytes, ram_backed_budget_bytes

    def need_bytes(per_rank):     # inputs + slots + outputs (four runs at a time) + FASTAs, all ranks, all RAM-backed
        return per_rank * world * (116000 + 4 * 16000 + 4 * 1500) + world * 5 * 4096 * SLOT_BYTES_PER_WINDOW

    def ram_bytes(per_rank):
        workers = world * e2e_generator_processes(world, cpus)
        return (need_bytes(per_rank) + workers * (per_rank / float(E2E_FILES_PER_RANK)) * E2E_GENERATOR_TRANSIENT
                + world * E2E_RANK_RESIDENT_BYTES)
    if free is None:
        free = ram_backed_budget_bytes()
        ram = ram_available_bytes() if ram is None else ram

    def fits(per_rank):
        return free > need_bytes(per_rank) * 1.1 and (ram is None or ram_bytes(per_rank) <= E2E_RAM_FRACTION * ram)
    n = windows_per_rank
    while may_shrink and n > 8192 and not fits(n):
        n = max(8192, n - 4096)
    where = "/dev/shm" if fits(n) else ("tmp" if world == 1 else None)
    return n, need_bytes(n), where, free


def plan_only(args, rank, world):
    """`--plan-only`: what this invocation WOULD do, decided without touching a device -- the legs, the call size, the
    end-to-end leg's size after the /dev/shm check, the files and worker processes of its input generation, and the host
    plan of its `world` ranks.  (A dry run of the driver's command shapes on a machine without the GPUs.)"""
    from helen_amd.host_plan import plan_host
    from helen_amd.synthetic import assembly_spec
    e2e_windows = (E2E_DEFAULT_WINDOWS if args.precision == "fp32" else 0) if args.e2e is None else args.e2e
    out = {"plan_only": True, "n_gpus": world, "rank": rank, "steps": args.steps, "warmup": args.warmup,
           "windows_per_step": args.batch * args.coalesce, "precision": args.precision,
           "legs": {"cpu_baseline": not args.no_cpu_baseline and world == 1, "host_path": not args.no_host_path,
                    "modes": not args.no_modes and args.precision == "fp32" and world == 1, "margins": not args.no_margins,
                    "end_to_end": e2e_windows > 0},
           "devices": "cuda:0 for every rank (--single-device)" if args.single_device else "cuda:LOCAL_RANK"}
    if e2e_windows > 0:
        n, need, where, free = e2e_size(e2e_windows, world, args.e2e is None)
        n_files = E2E_FILES_PER_RANK * world
        spec = assembly_spec(n * world, n_files)
        host = plan_host(list(range(world)), args.e2e_workers, 4096)
        out["end_to_end"] = {"windows_per_rank_asked": e2e_windows, "windows_per_rank": n, "shrunk": n != e2e_windows,
                             "ram_bytes_needed": need, "shm_free_bytes": free, "directory": where, "image_files": n_files,
                             "contigs": len(spec), "generation_processes_per_rank": e2e_generator_processes(world),
                             "stitch_threads": max(1, min(16, usable_cpus())), "host_plan": host.as_dict()}
    return out


def end_to_end(windows_per_rank, workers, batch, weights, rank, world, single_device, dist, may_shrink=False):
    """The product's commands over `world` ranks on a SIMULATED ASSEMBLY (helen_amd.synthetic.write_assembly_dir: contigs cut
    into 2400-position regions of three images, insert rows, short last images; pileups from the read-vote noise model;
    the TRAINED network of tests/golden/trained_synth.npz, so the called sequence is ~1 kb per image and neighbouring
    regions agree where they overlap), 16 image files per rank sharded round-robin (CallConsensusInterface.py:138-145):
      1. `call_consensus` -- image directory -> one prediction HDF5 per rank -- wall-clocked from the call to its return:
         host budgeting, process start-up, model load and the final close included (SURVEY.md 8d "end-to-end") = `value`;
      2. `polish` -- the same plus stitch -> FASTA (PolishInterface.py:49-105), stitch pipelined behind the inference
         (helen_amd/stitch_stream.py) = `polish_seconds`, and the FASTA compared with the two-phase stitch of the same
         prediction files.
    Every bench rank writes its share of the inputs (direct emitter of libhelen_io.so, RAM-backed directory); rank 0 then
    runs the commands, which start their OWN process per device exactly as the CLI does, while the other bench ranks
    sleep on a file (not in a collective: their GPUs must be idle).  Returns the `end_to_end` object on rank 0."""
    import shutil
    import tempfile

    from helen_amd import hdf5
    from helen_amd import predict as P
    from helen_amd.call_consensus import call_consensus, polish_genome
    from helen_amd.host_plan import ram_backed_budget_bytes
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import assembly_spec, write_assembly_dir
    box = [None, windows_per_rank, 0]
    # this rank's page-locked host-path buffers are still cached by torch's host allocator (3.3 GB): give them back
    import gc
    gc.collect()
    try:
        import torch
        torch.cuda.empty_cache()
        getattr(torch._C, "_host_emptyCache", lambda: None)()
    except Exception:           # noqa: BLE001 -- housekeeping only
        pass
    if rank == 0:
        box[1], box[2], where, free = e2e_size(windows_per_rank, world, may_shrink)
        if box[1] != windows_per_rank:
            sys.stderr.write("INFO: /dev/shm MAY TAKE %.1f GB (FREE SPACE, HALF OF THE AVAILABLE RAM): THE END-TO-END LEG SHRINKS TO "
                             "%d WINDOWS PER RANK.\n" % (free / 1e9, box[1]))
        if where is not None:
            box[0] = tempfile.mkdtemp(prefix="helen_e2e_", dir="/dev/shm" if where == "/dev/shm" else None)
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    d, windows_per_rank, need = box
    if d is None:
        return {"value": None, "skipped": "/dev/shm may take %.1f GB, the %d-rank leg needs %.1f GB"
                                          % (ram_backed_budget_bytes() / 1e9, world, need / 1e9)} if rank == 0 else None
    done = os.path.join(d, "done")
    try:
        img_dir = os.path.join(d, "img")
        os.makedirs(img_dir, exist_ok=True)
        n_files = E2E_FILES_PER_RANK * world
        spec = assembly_spec(windows_per_rank * world, n_files)
        t0 = time.time()
        write_error = None
        try:
            # file fi goes to caller fi % world (round-robin over the sorted list): this rank writes its own
            write_assembly_dir(img_dir, spec, n_files, direct=True, only_files=[fi for fi in range(n_files) if fi % world == rank],
                               processes=e2e_generator_processes(world))
        except Exception as e:          # noqa: BLE001 -- agreed on by all ranks below
            write_error = "%s: %s" % (type(e).__name__, e)
        t_write = time.time() - t0
        if dist is not None:            # doubles as the barrier: every rank learns whether all inputs exist
            write_error = next((e for e in gather_strings(dist, world, write_error or "") if e), None)
        if write_error:
            return {"value": None, "error": "writing the synthetic inputs failed: " + write_error} if rank == 0 else None
        if rank != 0:
            while not os.path.exists(done):
                time.sleep(0.05)
            return None
        model = os.path.join(d, "model.pkl")
        ModelHandler.save_model(weights, None, 128, 1, 0, model)
        out = os.path.join(d, "out")
        import contextlib
        device_ids = ",".join("0" if single_device else str(r) for r in range(world))
        threads = max(1, min(16, usable_cpus()))
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):     # the CLI prints the output file name: keep stdout to the JSON line
            call_consensus(img_dir, model, batch, workers, threads, out, "p", True, device_ids, world)
        dt = time.time() - t0
        run = dict(P.LAST_RUN)
        total = sum(r.get("windows", 0) for r in run.get("ranks", []))
        files = sorted(os.listdir(out))
        stored = 0
        for name in files:
            with hdf5.File(os.path.join(out, name)) as f:
                stored += sum(len(f.keys("predictions/" + c)) for c in f.keys("predictions"))
        # the two-phase stitch of those files (what `helen stitch` does, and what `polish` did until round 4) ...
        from helen_amd.stitch import perform_stitch
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):
            fasta = perform_stitch(out, os.path.join(d, "fa"), "asm", threads)
        dt_stitch = time.time() - t0
        # ... and the whole `polish` command's work in one go, stitch pipelined behind the inference
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):
            polish_genome(img_dir, model, batch, workers, threads, os.path.join(d, "polish"), "asm", True, device_ids, world)
        dt_polish = time.time() - t0
        polish_run = dict(P.LAST_RUN)
        fasta2 = os.path.join(d, "polish", "asm.fa")
        same = os.path.getsize(fasta) == os.path.getsize(fasta2)
        if same:
            with open(fasta, "rb") as a, open(fasta2, "rb") as b:
                while same:
                    x, y = a.read(1 << 24), b.read(1 << 24)
                    same = x == y
                    if not x:
                        break
        # ... and the COMMAND as a user starts it (one rank only: with several, the command would compete with this bench's
        # own ranks for their devices): `bin/helen polish -g` in a process of its own, from exec to exit -- interpreter
        # start, imports, device context, inference, stitch, FASTA, exit
        command = None
        if world == 1:
            import subprocess
            cmd = [sys.executable, os.path.join(ROOT, "bin", "helen"), "polish", "-i", img_dir, "-m", model, "-b", str(batch),
                   "-w", str(workers), "-t", str(threads), "-o", os.path.join(d, "cmd"), "-p", "asm", "-g"]
            t0 = time.time()
            r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            dt_cmd = time.time() - t0
            fasta3 = os.path.join(d, "cmd", "asm.fa")
            ok = r.returncode == 0 and os.path.isfile(fasta3) and os.path.getsize(fasta3) == os.path.getsize(fasta)
            if ok:
                with open(fasta, "rb") as a, open(fasta3, "rb") as b:
                    ok = a.read() == b.read()
            clock = [ln for ln in r.stderr.splitlines() if "WALL CLOCK" in ln]
            command = {"what": "bin/helen polish -i <dir> -m <model> -b %d -w %d -t %d -g, a process of its own, exec to exit"
                               % (batch, workers, threads),
                       "seconds": round(dt_cmd, 3), "windows_per_s": round(total / dt_cmd, 1), "returncode": r.returncode,
                       "fasta_equals_two_phase": bool(ok), "wall_clock_line": clock[-1][6:] if clock else None}
            # the same command in the two opt-in arithmetic modes: there the device is 1.5x / 6x faster and the HOST stages
            # (readers, writer, stitch) set the pace; the FASTA is compared with the fp32 one byte for byte (fp32x3: equal
            # on this assembly; bf16: labels differ at ~1e-5 of the positions)
            command["modes"] = {}
            for prec in ("fp32x3", "bf16"):
                out_dir = os.path.join(d, "cmd_" + prec)
                t0 = time.time()
                r2 = subprocess.run(cmd[:cmd.index("-o") + 1] + [out_dir] + cmd[cmd.index("-o") + 2:] + ["--precision", prec],
                                    cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                dt2 = time.time() - t0
                f2 = os.path.join(out_dir, "asm.fa")
                same2 = None
                if r2.returncode == 0 and os.path.isfile(f2):
                    same2 = os.path.getsize(f2) == os.path.getsize(fasta)
                    if same2:
                        with open(fasta, "rb") as a, open(f2, "rb") as b:
                            same2 = a.read() == b.read()
                busy = [ln for ln in r2.stderr.splitlines() if "WINDOWS IN" in ln]
                command["modes"][prec] = {"seconds": round(dt2, 3), "windows_per_s": round(total / dt2, 1),
                                          "returncode": r2.returncode, "fasta_equals_fp32": same2,
                                          "stages": busy[-1][6:busy[-1].index(";")] + ")" if busy and ";" in busy[-1] else None}
                shutil.rmtree(out_dir, ignore_errors=True)
        plan = run.get("host_plan", {})
        # the same run without its fixed costs: every rank's windows over the slowest rank's loop time (first slot
        # submitted .. last labels back; process start-up, model load, page-locking, file close and tear-down excluded)
        loops = [r["seconds"] - r["setup_seconds"] - r["close_seconds"] for r in run.get("ranks", [])
                 if all(r.get(k) is not None for k in ("seconds", "setup_seconds", "close_seconds"))]
        steady = round(total / max(loops), 1) if loops and max(loops) > 0 else None
        return {"value": round(total / dt, 1), "unit": "windows/s", "n_ranks": world, "windows": total,
                "seconds": round(dt, 3), "value_without_setup_and_close": steady,
                "weights": "trained_synth", "workload": "simulated assembly: %d contigs, %d regions of 2400 positions (three images "
                                                        "each, the last short), insert rows, %d image files" % (len(spec), stored, n_files),
                "usable_cpus": plan.get("usable_cpus"),
                "reader_workers_requested": workers,
                "reader_workers_per_rank": plan.get("reader_workers_per_rank"),
                "predicted_host_ceiling": plan.get("predicted_host_ceiling_windows_per_s"),
                "predicted_device_ceiling": plan.get("predicted_device_ceiling_windows_per_s"),
                "predicted_bound": plan.get("predicted_bound"),
                "per_rank": [{k: r.get(k) for k in ("rank", "device", "windows", "seconds", "stage_seconds",
                                                    "setup_seconds", "close_seconds", "reader_workers", "slots",
                                                    "cpus_pinned", "numa_node")} for r in run.get("ranks", [])],
                "host_plan_notes": plan.get("host_plan_notes") if "host_plan_notes" in plan else plan.get("notes"),
                "output_files": files, "regions_stored": stored,
                "stitch": {"seconds": round(dt_stitch, 3), "threads": threads, "fasta_bytes": os.path.getsize(fasta),
                           "what": "two-phase: perform_stitch on the finished prediction files"},
                "two_phase_polish_seconds": round(dt + dt_stitch, 3),
                "polish_seconds": round(dt_polish, 3),
                "polish_windows_per_s": round(total / dt_polish, 1),
                "polish": {"what": "polish_genome: call_consensus + stitch pipelined behind the inference (helen_amd/stitch_stream.py), "
                                   "image directory -> prediction HDF5 + FASTA",
                           "seconds": round(dt_polish, 3), "predict_seconds": polish_run.get("seconds"),
                           "fasta_equals_two_phase": bool(same),
                           "per_rank": [{k: r.get(k) for k in ("rank", "windows", "seconds", "stage_seconds", "stitch_stream")}
                                        for r in polish_run.get("ranks", [])],
                           # several ranks: the stitch collector processes (helen_amd/stitch_collect.py) -- regions, joins
                           # and seconds per collector, the parent's wait for them and its copy of their parts
                           "stitch_collectors": polish_run.get("stitch_collectors")},
                "polish_command": command,
                "what": "call_consensus(image_dir -> one prediction HDF5 per rank) over %d rank(s) incl. host "
                        "budgeting, process start-up, model load and close; %d synthetic windows per rank written in "
                        "%.1f s to %s" % (world, windows_per_rank, t_write, os.path.dirname(d))}
    finally:
        if rank == 0:
            open(done, "w").close()
            time.sleep(0.2)
            shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed device calls per rank")
    ap.add_argument("--warmup", type=int, default=2, help="untimed device calls per rank")
    ap.add_argument("--batch", type=int, default=256, help="windows per step (BASELINE.json: 256)")
    ap.add_argument("--coalesce", type=int, default=16, help="loader batches per device call")
    ap.add_argument("--mode", default="uniform", choices=["uniform", "pileup"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-margins", action="store_true", help="skip the label-margin report (random-init and peaked weights)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-memory -> host-memory leg")
    ap.add_argument("--no-modes", action="store_true",
                    help="skip the `modes` object (bf16 at batch 512 = BASELINE.json configs[3], and fp32x3, each under the "
                         "same clock with its own roofline and its label / logit check against the fp32 path)")
    ap.add_argument("--e2e", type=int, default=None, metavar="WINDOWS",
                    help="windows PER RANK of the end-to-end leg: the product's call_consensus over a synthetic HDF5 "
                         "image directory, N ranks under --gpus N (default %d: a bounded leg in every line; 300000 = "
                         "chr20 scale, needs ~120 KB of /dev/shm per window; 0 = off)" % E2E_DEFAULT_WINDOWS)
    ap.add_argument("--e2e-workers", type=int, default=8, help="readers per rank of the --e2e leg (-w of call_consensus)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp32x3"],
                    help="gate-matmul arithmetic; fp32 (true fp32 MFMA) is BASELINE.json configs[1], the "
                         "headline; fp32x3 = fp32-class via three-term bf16 splits (opt-in experiment)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-time reduce (nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="testing aid: every rank uses cuda:0 (exercise the N>1 code path on one GPU)")
    ap.add_argument("--no-traffic", action="store_true", help="(accepted and ignored: the counter passes are scripts/profile_round.sh's)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print what this invocation would do (legs, sizes, host plan of the N ranks) as one JSON line and "
                         "exit, without touching a device: a dry run of the launcher's command shapes")
    args = ap.parse_args()

    if args.plan_only:
        # under a launcher (WORLD_SIZE set) every rank reaches this point; rank 0 answers.  Started plainly with
        # --gpus N it answers for the N ranks it would become.
        world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
        if world != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
            sys.exit(2)
        rank = int(os.environ.get("RANK", "0"))
        if rank == 0:
            print(json.dumps(plan_only(args, rank, world)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become N ranks, one per GPU (predict_gpu.py:207-226 spawns one process
        # per device the same way)
        import socket
        import subprocess

        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not args.single_device:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, have))
            sys.exit(2)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
        sys.exit(2)
    if args.single_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d wants cuda:%d but %d GPU(s) are visible\n"
                         % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (any world size) the process group is used for the barrier and the
    # max-over-ranks time; a plain `python bench.py` needs none
    rccl = None          # how the RCCL leg of the barrier came up ("ok" / why not); None without a process group
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import datetime

        import torch.distributed as dist
        if args.single_device and os.environ.get("HELEN_BENCH_TRY_RCCL", "") != "1":
            args.dist_backend = "gloo"      # RCCL refuses two ranks on one device
        # The path has no data-path collective: the group only carries the barrier and the per-rank times.  Those
        # go over gloo (CPU tensors) whatever happens; with --dist-backend nccl the barrier ALSO crosses RCCL, whose
        # communicator is brought up here, guarded -- a node whose RCCL does not come up (IPC mode, duplicate
        # devices) still gives its scaling line, and the line says so.
        if args.dist_backend == "nccl":
            dist.init_process_group("cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=600))
            mine = 1
            try:
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce over RCCL returned %s for %d ranks" % (probe.item(), world))
            except Exception as e:      # noqa: BLE001 -- reported in the JSON line
                mine = 0
                rccl = "failed on rank %d: %s: %s" % (rank, type(e).__name__, str(e).splitlines()[0][:200])
            flag = torch.tensor([mine], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                rccl = "ok"
            elif rccl is None:
                rccl = "failed on another rank"
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
            rccl = "not used (--dist-backend gloo%s)" % ("; --single-device" if args.single_device else "")
    else:
        dist = None

    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_weights

    B, G = args.batch, args.coalesce
    call_windows = B * G
    eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=local_rank, max_windows=call_windows,
                      precision=args.precision)

    # synthetic chr20-scale image shard, resident in HBM before the timed region; every rank gets
    # its own shard (weak scaling: images are sharded by file, CallConsensusInterface.py:138-145)
    n_res = min(max(args.steps, args.warmup, 1), 8) * call_windows   # resident shard (re-walked if shorter)
    gen = torch.Generator(device=dev).manual_seed(20260928 + rank)
    if args.mode == "uniform":
        images = torch.randint(0, 256, (n_res, 1000, 90), dtype=torch.uint8, device=dev, generator=gen)
    else:
        images = torch.zeros((n_res, 1000, 90), dtype=torch.uint8, device=dev)
        cols = torch.randint(0, 90, (n_res, 1000, 4), device=dev, generator=gen)
        vals = torch.randint(1, 256, (n_res, 1000, 4), dtype=torch.uint8, device=dev, generator=gen)
        images.scatter_(2, cols, vals)
    bases = torch.empty((n_res, 1000), dtype=torch.uint8, device=dev)
    rles = torch.empty_like(bases)
    stream = torch.cuda.current_stream(dev).cuda_stream

    import ctypes

    from helen_amd import _lib
    lib = _lib.load()

    n_calls_res = n_res // call_windows

    def run(n_steps):
        for k in range(n_steps):
            s = (k % n_calls_res) * call_windows
            e = s + call_windows
            _lib.check(lib.helen_polish_batch(eng._handle, images[s:e].data_ptr(), call_windows,
                                              bases[s:e].data_ptr(), rles[s:e].data_ptr(), None, None,
                                              ctypes.c_void_p(stream)))

    token_dev = torch.zeros(1, device=dev) if rccl == "ok" else None
    token_cpu = torch.zeros(1)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            if token_dev is not None:           # over RCCL / xGMI
                dist.all_reduce(token_dev)
                torch.cuda.synchronize(dev)
            dist.all_reduce(token_cpu)          # over gloo: holds whatever RCCL does

    # HIP events around the dominant kernel only; the pairs are created during the warm-up (topped up to the
    # number of timed steps) and recycled afterwards: no event is created or destroyed inside the timed region
    eng.set_profiling(["gru_enc", "gru_dec"])
    run(args.warmup)
    if args.warmup < args.steps:   # untimed: as many event pairs as the timed steps will need
        run(args.steps - args.warmup)
    barrier()
    eng.reset_kernel_stats()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stats = eng.kernel_stats()
    # every kernel class of the call, bracketed launch by launch on the same stream, over a few extra calls OUTSIDE the timed
    # region (events around all 77 launches of a call would cost the headline ~0.3 %): `roofline.kernels`
    from helen_amd._lib import KERNEL_CLASSES
    table_calls = 4
    eng.set_profiling(list(KERNEL_CLASSES))
    run(2)
    torch.cuda.synchronize(dev)
    eng.reset_kernel_stats()
    run(table_calls)
    torch.cuda.synchronize(dev)
    all_stats = eng.kernel_stats()
    eng.set_profiling([])
    my_elapsed = elapsed
    clock = sustained_clock(run, dev) if world == 1 else None

    # host path: the same number of windows from page-locked host memory to labels in host memory, ONE
    # helen_polish_host call (it pipelines sub-batches of `call_windows`: upload k+1 | kernels k | download k-1)
    host_elapsed, hn, host_error = None, n_res, None
    if not args.no_host_path:
        # the resident shard (at most 8 device calls' worth), copied to page-locked host memory.  This leg must not
        # be able to take the headline down with it: a box that refuses to page-lock 3 GB reports the refusal.
        try:
            himg = torch.empty((hn, 1000, 90), dtype=torch.uint8).pin_memory()
            himg.copy_(images)
            hb = torch.empty((hn, 1000), dtype=torch.uint8).pin_memory()
            hr = torch.empty((hn, 1000), dtype=torch.uint8).pin_memory()
            eng.polish_host(himg[:call_windows], out=(hb.numpy()[:call_windows], hr.numpy()[:call_windows]))   # warm-up
        except Exception as e:          # noqa: BLE001 -- reported in the JSON line
            host_error = "%s: %s" % (type(e).__name__, e)
        barrier()
        if host_error is None:
            t0 = time.perf_counter()
            eng.polish_host(himg, out=(hb.numpy(), hr.numpy()))
        barrier()
        if host_error is None:
            host_elapsed = time.perf_counter() - t0
            # the host path must give the labels of the device path (which has walked the whole resident shard
            # when steps >= its length in calls)
            k = min(args.steps, n_calls_res) * call_windows
            if not (torch.equal(hb[:k], bases[:k].cpu()) and torch.equal(hr[:k], rles[:k].cpu())):
                sys.stderr.write("bench.py: host-path labels differ from the device-path labels\n")
                sys.exit(3)
            del himg, hb, hr

    per_rank = [my_elapsed]
    if dist is not None:
        t = torch.tensor([elapsed, host_elapsed or 0.0], dtype=torch.float64)       # (CPU tensors: gloo)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [float(e[0].item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if host_elapsed is not None:
            host_elapsed = float(t[1].item())

    if rank == 0:
        total_windows = world * args.steps * call_windows
        value = total_windows / elapsed
        gru_ms = stats["gru_enc"][0] + stats["gru_dec"][0]
        gru_n = stats["gru_enc"][1] + stats["gru_dec"][1]
        avg_ms = gru_ms / max(gru_n, 1)
        win_per_launch = call_windows
        achieved = GRU_FLOP_PER_WINDOW_LAUNCH * win_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(win_per_launch, args.precision)
        # (A counter pass cannot run inside this process's run: rocprofv3 --pmc is collected in passes of its own, one
        # counter each, with nothing else on the device -- scripts/profile_round.sh does that and scripts/pmc_summary.py
        # writes the committed summary these figures are read from.)
        call_bytes = pmc_call_bytes_per_window(args.precision)
        peak = BF16_MFMA_PEAK if args.precision == "bf16" else FP32_MFMA_PEAK
        bound, unit = "mfma", "TFLOP/s"
        if args.precision == "bf16":
            # the fused layer kernels do projection + recurrence (+ the decoder's head partials): count a
            # layer launch's algorithmic matmul FLOPs, averaged over the encoder and decoder launches
            flop = BF16_LAYER_FLOP_PER_WINDOW_LAUNCH
            achieved = flop * win_per_launch / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": "pileup windows/sec (batch 256, 1000-pos)",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world, "ranks_seen": len(per_rank),
            "per_rank_windows_per_s": [round(args.steps * call_windows / t, 1) for t in per_rank],
            "devices": "cuda:0 for every rank (--single-device)" if args.single_device else "cuda:LOCAL_RANK",
            "barrier": None if dist is None else ("RCCL all-reduce + gloo all-reduce" if rccl == "ok" else "gloo all-reduce"),
            "rccl": rccl,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 operands, f32 accumulate/state",
                      "fp32x3": "f32 emulated as 3-term bf16 splits (6 exact partial products), f32 "
                                "accumulate/state"}[args.precision],
            "data": "synthetic (%s uint8 windows, seeded; random-init weights of the reference "
                    "architecture)" % args.mode,
            "config": {"workload": ("BASELINE.json configs[1]: 1xMI355X, batch 256, fp32, synthetic "
                                    "chr20-scale image shard resident in HBM") if args.precision != "bf16"
                       else "BASELINE.json configs[3] variant: bf16 gate matmuls, fp32 accumulate/state",
                       "batch": B, "batches_per_step": G, "windows_per_step": call_windows,
                       "arithmetic": {"fp32": "fp32 operands, products and accumulation on v_mfma_f32_16x16x4_f32 (recurrences, decoder "
                                              "projection, heads); the encoder's input projection takes EXACT bf16 products -- a pileup "
                                              "count is one bf16 term, W_ih three -- with fp32 accumulation (roofline.kernels names the "
                                              "pipe of every kernel class)",
                                      "bf16": "gate-matmul operands rounded to bf16, fp32 accumulation, state and gate math",
                                      "fp32x3": "every fp32 product as the six leading products of three-term bf16 splits, fp32 "
                                                "accumulation, state and gate math"}[args.precision],
                       "positions": 1000, "features": 90, "windows_per_gpu": args.steps * call_windows,
                       "resident_windows_per_gpu": n_res, "sharding": "by rank, no collective"},
            "roofline": {"bound": bound, "kernel": {"fp32": "gru_pair_kernel (GRU recurrence of two window tiles per 8-wave workgroup, fp32 MFMA; "
                                            "decoder launches include the heads' product)",
                                    "bf16": "gru_fused_bf16_il_kernel (encoder and decoder launches): projection + recurrence per layer, "
                                            "two window tiles per workgroup, bf16 MFMA; bound in practice by the fp32 gate math, not the matrix pipe",
                                    "fp32x3": "gru_x3_kernel (GRU recurrence, 6 bf16 MFMAs per fp32 product "
                                              "group; fraction is of the fp32 MFMA peak)"}[args.precision],
                         "achieved": round(achieved, 2),
                         "peak": peak / (1e12 if bound == "mfma" else 1e9), "unit": unit,
                         "frac": round(achieved * (1e12 if bound == "mfma" else 1e9) / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_measured_in_this_run": False if traffic is not None else None,
                         # the whole call's HBM traffic per window against the 92,000 algorithmic bytes, from the
                         # same committed counter passes: what staging gi / y1 between the kernels costs (the path is
                         # MFMA-bound, so it is power, not time)
                         "hbm_bytes_per_window_from_profile": call_bytes,
                         "hbm_over_algorithmic": None if not call_bytes else round(call_bytes / 92000.0, 1),
                         "avg_launch_ms": round(avg_ms, 4), "launches": gru_n,
                         "avg_launch_ms_encoder": round(stats["gru_enc"][0] / max(stats["gru_enc"][1], 1), 4),
                         "avg_launch_ms_decoder": round(stats["gru_dec"][0] / max(stats["gru_dec"][1], 1), 4),
                         "path_frac": round(value / world * FLOP_PER_WINDOW /
                                            (BF16_MFMA_PEAK if args.precision == "bf16" else FP32_MFMA_PEAK), 4),
                         "kernels": kernel_table(all_stats, table_calls, call_windows, args.precision, my_elapsed * 1e3 / args.steps)},
        }
        if clock is not None:
            with_clock(out["roofline"], clock)
        if args.precision != "fp32":
            out["precision_check"] = precision_check(eng, args.precision, images, dev)
        if not args.no_margins:
            eng.close()            # the reports build engines of their own
            if args.precision == "fp32":
                try:
                    out["call_sizes"] = call_size_report(images, dev)
                except Exception as e:      # noqa: BLE001 -- reported in the JSON line
                    out["call_sizes"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["margins"] = margin_report(images, dev, args.precision)
        if host_error is not None:
            out["host_path"] = {"value": None, "error": host_error}
        if host_elapsed is not None:
            hv = world * hn / host_elapsed
            out["host_path"] = {
                "value": round(hv, 1), "unit": "windows/s", "vs_device_resident": round(hv / value, 4),
                "what": "page-locked host uint8 images -> labels in page-locked host memory, one helen_polish_host "
                        "call per rank over %d windows (sub-batches of %d: upload k+1 | kernels k | download k-1); "
                        "PCIe included; labels checked equal to the device path" % (hn, call_windows),
                "h2d_GBps": round(hv / world * 90000 / 1e9, 2)}
        if not args.no_modes and args.precision == "fp32" and world == 1:
            eng.close()
            out["modes"] = {}
            for prec, mb in (("bf16", 512), ("fp32x3", B)):
                try:
                    out["modes"][prec] = mode_report(prec, mb, G, images, dev, (bases, rles))
                except Exception as e:      # noqa: BLE001 -- a mode must not take the headline down with it
                    out["modes"][prec] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B)
            try:
                out["cpu_baseline_torch_eager"] = torch_eager_baseline(B)
            except Exception as e:          # noqa: BLE001 -- an extra leg must not take the headline down with it
                out["cpu_baseline_torch_eager"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            try:
                out["host_mode"] = host_mode()
            except Exception as e:          # noqa: BLE001 -- an extra leg must not take the headline down with it
                out["host_mode"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
    eng.close()                # call_consensus builds its own engines (15.9 GB of scratch each)
    # (call_consensus is the fp32 product path: the opt-in arithmetic modes carry the leg only on request)
    e2e_windows = (E2E_DEFAULT_WINDOWS if args.precision == "fp32" else 0) if args.e2e is None else args.e2e
    e2e = None
    if e2e_windows > 0:
        try:
            e2e = end_to_end(e2e_windows, args.e2e_workers, B, trained_weights(), rank, world,
                             args.single_device, dist, may_shrink=args.e2e is None)
        except (Exception, SystemExit) as e:    # noqa: BLE001 -- this leg must not take the headline down with it (call_consensus exits on bad input)
            e2e = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if e2e is not None:
            out["end_to_end"] = e2e
        print(json.dumps(out))
    if dist is not None:
        dist.all_reduce(token_cpu)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
