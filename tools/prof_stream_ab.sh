cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 31; do
  MD_CONV_VARIANT=$v rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v$v -o p -- python bench.py --steps 10 --warmup 3 > gpurun_out/prof_v$v.log 2>&1 || exit 1
done
ls gpurun_out/prof_v0 gpurun_out/prof_v31
