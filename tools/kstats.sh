#!/bin/bash
# kernel-trace summary of one headline step (run through gpurun): tools/kstats.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-ks}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_kt.log 2>&1 || exit 2
cd $R
python3 tools/rocpd_stats.py $(find $OUT/${TAG}_kt -name "*.db" | head -1) $OUT/${TAG}_kernel_stats.txt > /dev/null
rm -rf $OUT/${TAG}_kt
head -24 $OUT/${TAG}_kernel_stats.txt | cut -c1-140
