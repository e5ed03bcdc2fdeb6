"""Developer probe (CPU): the stitch work of an N-RANK `polish` on a simulated assembly whose regions are dealt over the
ranks in blocks of `block` consecutive regions (MarginPolish scatters a contig over its threads' files), two ways:
  streams     every rank keeps and joins its own regions, saves them, the parent loads all and finishes (round 5's first form);
  collectors  every rank exports its regions to collector processes sharded by contig (helen_amd/stitch_collect.py).
Regions are produced as fast as the host can (no device in the loop), so the figures are the stitch stage's own throughput:
what matters on a node is whether it keeps up with ranks x 27 k regions/s and what is left after the last region.
    python scripts/dev/stitch_tail_bench.py [regions=200000] [ranks=8] [threads=8] [block=8] [pace=0]
pace = regions per second the "ranks" deliver in the collectors leg (0 = as fast as the host can)."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def regions_of(n, contigs, ranks, block, rng):
    """-> per contig: per rank (keys, list of sequences)."""
    per = n // contigs
    for c in range(contigs):
        truth = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, per * 3300 + 1000)].tobytes()
        out = [([], []) for _ in range(ranks)]
        for k in range(per):
            r = (k // block) % ranks
            s = k * 3300
            out[r][0].append(("contig_%02d" % c, s, s + 4000))
            out[r][1].append(truth[s:s + 4000])
        yield out


def main():
    from helen_amd import stitch_collect, stitch_stream
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    block = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    pace = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    contigs = 32
    d = tempfile.mkdtemp(prefix="tail_", dir="/dev/shm")
    try:
        pred = os.path.join(d, "pred")
        os.makedirs(pred)
        files = [os.path.join(pred, "p_%d.hdf" % r) for r in range(ranks)]
        for f in files:
            open(f, "wb").close()
        data = list(regions_of(n, contigs, ranks, block, np.random.default_rng(5)))
        # ---- streams ----
        t0 = time.time()
        streams = [stitch_stream.RegionStream(f, max(1, threads // ranks)) for f in files]
        for out in data:
            for r in range(ranks):
                if out[r][0]:
                    streams[r].accept_sequences(out[r][0], out[r][1])
        res = [s.finish() for s in streams]
        t1 = time.time()
        paths = [x.save(d) for x in res]
        t2 = time.time()
        del res, streams
        res = [stitch_stream.StreamResult.load(p) for p in paths]
        t3 = time.time()
        a = stitch_stream.finish_stitch(res, pred, os.path.join(d, "a"), "asm", threads)
        t4 = time.time()
        print("streams:    ranks' own joins %.1f s (in the ranks, behind the devices) | save %.2f s (in the ranks) | AFTER THE LAST "
              "REGION, in the parent: load %.2f s + finish %.2f s = %.2f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t2), flush=True)
        del res
        # ---- collectors ----
        run = stitch_collect.CollectorRun(files, threads, directory=d).start()
        t0 = time.time()
        exports = [stitch_stream.RegionStream(f, 1, export=stitch_collect.RegionExport(run.export_spec()[0], r, run.export_spec()[1]))
                   for r, f in enumerate(files)]
        sent = 0
        for out in data:
            for r in range(ranks):
                if out[r][0]:
                    exports[r].accept_sequences(out[r][0], out[r][1])
                    sent += len(out[r][0])
            if pace > 0:
                time.sleep(max(0.0, sent / pace - (time.time() - t0)))
        for e in exports:
            e.finish()
        t1 = time.time()
        b = run.finish(os.path.join(d, "b"), "asm")
        t2 = time.time()
        print("collectors: %d collector(s); export of %d regions %.2f s; AFTER THE LAST REGION: %.2f s (all regions arrived at once: "
              "nothing was hidden)" % (run.buckets, n, t1 - t0, t2 - t1))
        print("            per collector:", [(s["regions"], s["seconds"], s["late"], s["sliced"], s["contigs"], s["aligned_now"]) for s in run.stats["per_collector"]])
        print("same FASTA:", open(a, "rb").read() == open(b, "rb").read())
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
