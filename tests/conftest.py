import faulthandler
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# One hang must not erase the evidence of a whole run (round 3: a subprocess that never exited took 135 GPU tests with
# it).  Three independent guards:
#   1. every subprocess a test starts runs in a process group of its own, under a capped timeout, and the WHOLE group is
#      killed on expiry (subprocess.run's own timeout kills only the direct child, not a launcher's workers);
#   2. a per-test watchdog (SIGALRM): stacks of all threads to stderr, then the test FAILS and the session goes on;
#   3. collection order: the HIP-vs-oracle parity files first, the subprocess / bench files last.
SUBPROCESS_CAP_S = float(os.environ.get("TP_TEST_SUBPROCESS_CAP", "240"))
SUBPROCESS_CEILING_S = float(os.environ.get("TP_TEST_SUBPROCESS_CEILING", "900"))   # upper bound of any explicit timeout
PER_TEST_LIMIT_S = int(os.environ.get("TP_TEST_LIMIT", "300"))

_ORDER = ["test_harness", "test_gpu_parity", "test_golden", "test_gpu_configs", "test_gpu_fine_generations", "test_gpu_refksp", "test_mma",
          "test_abi", "test_oracle_elements", "test_oracle_filter", "test_oracle_solver", "test_oracle_refksp", "test_mpiio",
          "test_cpp_host", "test_reference_on_shim", "test_multirank", "test_bench_line"]

_plain_run = subprocess.run


def hardened_run(cmd, *args, timeout=None, input=None, **kw):
    """subprocess.run with a process group per child and a killpg on expiry.  Semantics of subprocess.run are kept:
    an explicit `timeout` is honoured up to SUBPROCESS_CEILING_S -- the cap SUBPROCESS_CAP_S only applies to calls that pass none --, expiry
    raises subprocess.TimeoutExpired (after the group is dead; .output / .stderr carry what was captured, and the tail is
    printed so that an uncaught expiry still shows where the child was), `input=` goes through communicate()."""
    # an explicit timeout is honoured up to a global ceiling (one test with a huge timeout must not stall the suite)
    limit = min(timeout, SUBPROCESS_CEILING_S) if timeout is not None else SUBPROCESS_CAP_S
    capture = kw.pop("capture_output", False)
    check = kw.pop("check", False)
    if capture:
        kw["stdout"], kw["stderr"] = subprocess.PIPE, subprocess.PIPE
    if input is not None:
        kw["stdin"] = subprocess.PIPE
    kw["start_new_session"] = True
    p = subprocess.Popen(cmd, *args, **kw)
    try:
        out, err = p.communicate(input=input, timeout=limit)
    except subprocess.TimeoutExpired as e:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        try:
            out, err = p.communicate(timeout=15)
        except Exception:
            out, err = None, None
        tail = lambda b: (b if isinstance(b, str) else (b or b"").decode(errors="replace"))[-3000:]
        sys.stderr.write("subprocess %r did not finish within %.0f s; its process group was killed\n---- stdout ----\n%s\n---- stderr ----\n%s\n"
                         % (cmd, limit, tail(out), tail(err)))
        raise subprocess.TimeoutExpired(cmd, limit, output=out, stderr=err) from e
    r = subprocess.CompletedProcess(cmd, p.returncode, out, err)
    if check:
        r.check_returncode()
    return r


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    subprocess.run = hardened_run


def pytest_unconfigure(config):
    subprocess.run = _plain_run


def pytest_collection_modifyitems(config, items):
    def key(it):
        name = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 3   # unknown files: before the subprocess-heavy ones
    items.sort(key=key)     # stable: the order inside a file is kept


class _TestTimeout(Exception):
    pass


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    if not hasattr(signal, "SIGALRM"):
        yield
        return
    t0 = time.time()

    def on_alarm(signum, frame):
        sys.stderr.write("\n==== %s exceeded %d s: stacks of all threads ====\n" % (item.nodeid, PER_TEST_LIMIT_S))
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        raise _TestTimeout("%s exceeded the per-test limit of %d s (%.0f s)" % (item.nodeid, PER_TEST_LIMIT_S, time.time() - t0))

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(PER_TEST_LIMIT_S)
    try:
        yield
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle
