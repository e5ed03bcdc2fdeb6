"""Process targets of tests/test_host_plan.py (spawned: they must be importable by module path)."""
import os
import signal
import sys
import time


def sleeper(rank, marker_dir, result_q):
    """A rank that would run for a minute; SIGTERM makes it unwind (its `finally` leaves a marker)."""
    def on_term(signum, frame):
        raise SystemExit(143)
    signal.signal(signal.SIGTERM, on_term)
    try:
        os.setpgid(0, 0)
    except OSError:
        pass
    try:
        open(os.path.join(marker_dir, "started_%d" % rank), "w").close()
        time.sleep(60)
        result_q.put((rank, {"slept": True}))
    finally:
        open(os.path.join(marker_dir, "unwound_%d" % rank), "w").close()


def stubborn(rank, marker_dir, result_q):
    """A rank that ignores SIGTERM: only the group kill ends it."""
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    try:
        os.setpgid(0, 0)
    except OSError:
        pass
    open(os.path.join(marker_dir, "started_%d" % rank), "w").close()
    time.sleep(60)


def failing(rank, marker_dir, result_q, wait_for=(0,)):
    """Exits with 3 once every rank in `wait_for` is up (its handlers installed: a sibling that is terminated while it is
    still importing dies of the signal without unwinding, which is not what the tests are about)."""
    deadline = time.time() + 20
    while time.time() < deadline and not all(os.path.exists(os.path.join(marker_dir, "started_%d" % r))
                                             for r in wait_for):
        time.sleep(0.05)
    sys.exit(3)


def quick(rank, marker_dir, result_q):
    result_q.put((rank, {"rank": rank, "pid": os.getpid()}))


def sleeper_entry(rank, marker_dir, result_q):
    sleeper(rank, marker_dir, result_q)


def parent_main(marker_dir):
    """A parent that runs two sleeping ranks through helen_amd.predict.run_ranks (the test signals THIS process)."""
    from helen_amd.predict import run_ranks
    open(os.path.join(marker_dir, "parent_%d" % os.getpid()), "w").close()
    run_ranks(sleeper_entry, [(0, marker_dir), (1, marker_dir)], grace_seconds=5.0)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    parent_main(sys.argv[1])
