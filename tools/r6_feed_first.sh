#!/bin/bash
out=gpurun_out/${1:-r6h}
mkdir -p $out
cd /root/repo
for rep in 1 2; do
for v in "XH_RSUM_SEAL=1 XH_FEED_FIRST=192" "XH_RSUM_SEAL=1 XH_FEED_FIRST=256" "XH_RSUM_SEAL=0"; do
  echo "== $v" >> $out/fed_penalty2.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
k=r['kernels']
print('ms_per_step %.3f  route in step %.3f alone %.3f  pm in step %.3f abcd_sim in step %.3f' % (r['ms_per_step'], k['mrtm_route']['avg_ms'], k['mrtm_route'].get('avg_ms_alone',0), k['pm_pet'].get('avg_ms_in_fed_step',0), k['abcd_sim'].get('avg_ms_in_fed_step',0)))
" >> $out/fed_penalty2.txt
done
done
cat $out/fed_penalty2.txt
