"""The suite's own safety net (tests/conftest.py): a subprocess that outlives its timeout is killed WITH its children, and the
test that started it fails instead of hanging the run (round 3: one subprocess that never exited took 135 GPU tests with it)."""
import os
import subprocess
import sys
import time

import pytest


def test_a_hanging_subprocess_is_killed_with_its_children(tmp_path):
    pidfile = tmp_path / "grandchild.pid"
    # a launcher-like parent whose worker (grandchild) would survive a plain kill of the parent
    code = ("import subprocess, sys, time\n"
            "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
            "open(%r, 'w').write(str(p.pid))\n"
            "time.sleep(600)\n") % str(pidfile)
    t0 = time.time()
    with pytest.raises(subprocess.TimeoutExpired) as ei:   # subprocess.run's own exception: callers' handlers keep working
        subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=2)   # (conftest's hardened run)
    assert ei.value.timeout == 2 and time.time() - t0 < 30
    pid = int(pidfile.read_text())
    for _ in range(50):   # the grandchild is gone (or a zombie being reaped), not sleeping on
        try:
            state = open("/proc/%d/stat" % pid).read().split()[2]
        except OSError:
            state = None
        if state in (None, "Z", "X"):
            break
        time.sleep(0.1)
    assert state in (None, "Z", "X"), state


def test_subprocess_results_pass_through_unchanged():
    r = subprocess.run([sys.executable, "-c", "import sys; print('out'); print('err', file=sys.stderr); sys.exit(3)"],
                       capture_output=True, text=True, timeout=30)
    assert r.returncode == 3 and r.stdout == "out\n" and r.stderr == "err\n"
    with pytest.raises(subprocess.CalledProcessError):
        subprocess.run([sys.executable, "-c", "raise SystemExit(2)"], check=True, timeout=30)


def test_input_goes_through_the_hardened_run_too():
    r = subprocess.run([sys.executable, "-c", "import sys; print(sys.stdin.read().upper())"], input="abc", capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and r.stdout == "ABC\n"
    with pytest.raises(subprocess.TimeoutExpired):
        subprocess.run([sys.executable, "-c", "import sys, time; sys.stdin.read(); time.sleep(600)"], input="x", capture_output=True, text=True, timeout=2)


def test_collection_order_puts_parity_first_and_subprocess_files_last(request):
    from tests import conftest as c   # noqa: F401  (importable as a module: the order list is data)
    order = c._ORDER
    assert order.index("test_gpu_parity") < order.index("test_cpp_host") < order.index("test_multirank") < order.index("test_bench_line")
    assert order[-1] == "test_bench_line"
