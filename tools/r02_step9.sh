#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02i; mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest.log
