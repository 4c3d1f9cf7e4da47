#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python3 tests/diag/gpu_fill_compare.py 2>&1 | grep -v amdgpu.ids | tail -30
