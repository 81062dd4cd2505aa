#!/bin/bash
# round 6, trunk-by-XCD for unequal trunks: the README configuration's frame (view-direction static trunk, 23 %
# longer than the dynamic one) on this tree and on the round-5 tree (base_r05/: `git archive` of the round-5 commit, built), the two
# interleaved on ONE box; then FETCH_SIZE / WRITE_SIZE / TCC hit counters of the field launches of this tree.
#   usage: bash tools/gpu/r06_xcd.sh <tag>
TAG=${1:-a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2 3; do
  for tree in "$ROOT" "$ROOT/base_r05"; do
    echo "== round $rnd  tree $(basename $tree)" >> $O/xcd_ab.txt
    python $tree/tools/debug/readme_frame_timing.py 5 2>&1 | grep "side rows" >> $O/xcd_ab.txt
  done
done
cat $O/xcd_ab.txt
pmc() { local name=$1 tree=$2; shift 2
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/readme_pmc_$name -o $name -- python $tree/tools/debug/readme_frame_timing.py 1 > $O/readme_pmc_$name.log 2>&1
}
pmc fetch $ROOT FETCH_SIZE
pmc write $ROOT WRITE_SIZE
pmc tcc $ROOT TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc fetch_base $ROOT/base_r05 FETCH_SIZE
python $ROOT/profiles/summarize_r06_xcd.py $O > $O/xcd_pmc_summary.txt 2>&1
cat $O/xcd_pmc_summary.txt
# raw traces stay here
find $O -name "*.csv" -size +2M -delete
