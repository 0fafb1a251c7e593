#!/bin/bash
# round 6 A/Bs on one box: (1) PM class block through LDS, (2) stream ring size x check interval (item 5)
out=gpurun_out/${1:-r6k}
mkdir -p $out
cd /root/repo
export XH_CACHE_DIR=/tmp/xh_cache_ab
L=xanthos_amd
echo "# tools/pm_ab.py: base library against -DXH_PM_CLASS_LDS=1 (per-class vectors read from LDS) and =2 (next class's block read ahead into registers); config 2, 20 steps, alternating" > $out/pm_class_lds_ab.txt
timeout 900 python tools/pm_ab.py $L/libxanthos_hip.so $L/libxanthos_hip_pmlds1.so >> $out/pm_class_lds_ab.txt 2>&1
timeout 900 python tools/pm_ab.py $L/libxanthos_hip.so $L/libxanthos_hip_pmlds2.so >> $out/pm_class_lds_ab.txt 2>&1
echo "# XH_FLOW_RS (sub-steps per stream ring) x check interval CH (sub-steps between flow-control checks; 128: make exprsum EXPFLAGS=-DXH_WAVE_CH=128), tools/rsum_probe.py 600 120 3" > $out/ring_ch_sweep.txt
for rep in 1 2; do
for lib in libxanthos_hip.so libxanthos_hip_ch128.so; do
  for rs in 4096 8192 16384; do
    echo "== $lib XH_FLOW_RS=$rs" >> $out/ring_ch_sweep.txt
    XH_LIBRARY=$PWD/$L/$lib XH_FLOW_RS=$rs timeout 600 python tools/rsum_probe.py 600 120 3 2>&1 | grep -E "PARITY|reassoc  mrtm_route|cycles per sub-step|imports|exports|both|pair units" >> $out/ring_ch_sweep.txt
  done
done
done
cat $out/pm_class_lds_ab.txt; grep -E "^==|mrtm_route" $out/ring_ch_sweep.txt
