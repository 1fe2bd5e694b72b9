#!/bin/bash
set -u
O=gpurun_out/r04c; mkdir -p $O
cd "$(dirname "$0")/.."
timeout 600 python tools/dbg_shape.py 13312 3328 4992 bulk_bn=128 skip_zeros=0 group=4 fuse_forward=0 sweep_persist=0 lookahead=0 > $O/dbg_shape.txt 2>&1
cat $O/dbg_shape.txt
