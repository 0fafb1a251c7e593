#!/bin/bash
# round 6: single-sum plans on the GPU box: placement / halo / import-cap variants
out=gpurun_out/${1:-r6b}
mkdir -p $out
cd /root/repo
export XH_CACHE_DIR=/tmp/xh_cache
run() {
  echo "== $*" >> $out/probe.txt
  env "$@" XH_FLOW_CHECK=1 timeout 600 python tools/rsum_probe.py 600 120 3 >> $out/probe.txt 2>&1
}
run XH_RSUM_PAIR_IMPORTS=8
run XH_RSUM_EXCL=1
run XH_RSUM_PAIR_IMPORTS=8
run XH_RSUM_PAIR_IMPORTS=8 XH_RSUM_HALO=6
