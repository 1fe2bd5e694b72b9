#!/bin/bash
O=gpurun_out/r06s; mkdir -p $O
for i in 1 2 3; do ( timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "exchange_forms or wide_bulk or stall or ladder" -p no:cacheprovider > $O/dist_$i.log 2>&1; echo "rc=$?" >> $O/dist_$i.log ); tail -2 $O/dist_$i.log; done
( timeout 600 python tools/chain_stress.py 300 7 > $O/chain_stress.txt 2>&1 ); tail -1 $O/chain_stress.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
