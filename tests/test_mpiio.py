"""Output container (MPIIO.cc) and restart files: round trip with an independent reader, and -- where the
reference checkout exists (build container only) -- conversion by the reference's own bin2vtu_v3.py."""
import base64
import os
import re
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from topopt_in_petsc_amd.mpiio import MPIIO, read_output, write_restart, read_restart
from topopt_in_petsc_amd.partition import SlabPartition

REF = "/root/reference"


def _write(tmp, ndumps=2):
    part = SlabPartition(5, 4, 3)
    h = 0.5
    fn = os.path.join(tmp, "output_00000.dat")
    io = MPIIO(part, h, filename=fn)
    rng = np.random.default_rng(0)
    dumps = []
    for it in range(1, ndumps + 1):
        U = rng.standard_normal(3 * part.n_local_nodes)
        x, xt, xp = rng.random(part.n_own_elems), rng.random(part.n_own_elems), rng.random(part.n_own_elems)
        io.WriteVTK(U, x, xt, xp, it)
        dumps.append((it, U, x, xt, xp))
    return part, h, fn, dumps


def test_container_round_trip(tmp_path):
    part, h, fn, dumps = _write(str(tmp_path))
    d = read_output(fn)
    assert d["info"] == "TopOpt result version 1.1\n"
    assert d["pnames"] == "ux, uy, uz" and d["cnames"] == "x, xTilde, xPhys"
    assert d["points"].shape == (60, 3) and d["conn"].shape == (24, 8)
    assert np.allclose(d["points"][-1], [2.0, 1.5, 1.0])
    assert (d["types"] == 12).all() and np.array_equal(d["offsets"], 8 * np.arange(1, 25))
    # hexahedra: node 6 is the corner opposite to node 0
    p = d["points"]
    assert np.allclose(p[d["conn"][:, 6]] - p[d["conn"][:, 0]], h)
    assert len(d["dumps"]) == 2
    for (it, pf, cf), (it0, U, x, xt, xp) in zip(d["dumps"], dumps):
        assert it == it0
        assert np.array_equal(pf, U.reshape(-1, 3).T.astype(np.float32))
        assert np.array_equal(cf, np.stack([x, xt, xp]).astype(np.float32))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "bin2vtu_v3.py")), reason="reference checkout not present")
def test_reference_converter_accepts_the_file(tmp_path):
    """the reference's own post-processor (bin2vtu_v3.py + makevtu_v3.py, python stdlib only) parses our file"""
    part, h, fn, dumps = _write(str(tmp_path), ndumps=3)
    env = dict(os.environ, PYTHONPATH=REF)
    r = subprocess.run([sys.executable, os.path.join(REF, "bin2vtu_v3.py"), "1"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Done" in r.stdout, r.stdout + r.stderr
    vtu = os.path.join(str(tmp_path), "output_00001.vtu")
    txt = open(vtu, "rb").read()
    assert b'NumberOfPoints="60" NumberOfCells="24"' in txt
    # dataset 1 = second dump: decode the last cell field (xPhys) from the base64 payload
    arrays = re.findall(rb'<DataArray[^>]*Name="([^"]*)"[^>]*>\s*([A-Za-z0-9+/=]+)', txt)   # header and data: two base64 runs
    names = [a[0].decode() for a in arrays]
    # 3 mesh arrays + 6 fields; the converter's readInString drops the last character of each name list
    # (bin2vtu_v3.py:153), so the reference's own files also show "u" / "xPhy" -- the bytes written are the reference's
    assert names[:3] == ["connectivity", "offsets", "types"] and names[3:] == ["ux", "uy", "u", "x", "xTilde", "xPhy"]
    payload = arrays[-1][1]
    n, = struct.unpack("<Q", base64.b64decode(payload[:12]))
    vals = np.frombuffer(base64.b64decode(payload[12:]), dtype="<f4")
    assert n == 4 * vals.size
    assert np.array_equal(vals, dumps[1][4].astype(np.float32))


def test_restart_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    vs = [rng.random(17) for _ in range(4)] + [rng.random(51), rng.random(17)]
    pre = os.path.join(str(tmp_path), "Restart00")
    write_restart(pre, 37, 0.0928, *vs)
    itr, fscale, *back = read_restart(pre)
    assert itr == 37 and fscale == pytest.approx(0.0928)
    for a, b in zip(vs, back):
        assert np.array_equal(a, b)
    raw = open(pre + ".dat", "rb").read()
    assert struct.unpack(">ii", raw[:8]) == (1211214, 17)   # big-endian class id + length, then big-endian doubles
