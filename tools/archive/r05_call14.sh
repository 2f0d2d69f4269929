#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "arbiter_on_the_operator" 2>&1 | tail -15) 2>&1 | tail -20
