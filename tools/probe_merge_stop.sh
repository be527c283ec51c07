#!/bin/bash
# diagnostic: rocprof duration of merge_hybrid_kernel when it returns after phase N (ablation build)
export VODHIP_LIB=$PWD/vod_amd/csrc/libvodhip_ablation.so
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
for stop in 0 1 2 3 4 5 6 -1; do
  export VODHIP_HY_STOP=$stop
  [ $stop = -1 ] && unset VODHIP_HY_STOP
  rm -rf /tmp/pm_$stop
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_$stop -- python3 $ROOTD/tools/probe_c5.py > /tmp/pm_$stop.log 2>&1
  f=$(find /tmp/pm_$stop -name "*kernel_stats.csv" | head -1)
  echo "stop=$stop $(grep merge_hybrid $f | cut -d, -f2-6 | tr -d '"' | sed 's/^.*int),//')"
done
