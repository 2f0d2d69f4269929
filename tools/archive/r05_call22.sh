#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -25) 2>&1 | tail -32
