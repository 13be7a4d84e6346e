export TMPDIR=/tmp; cd /tmp
R=$1
for v in 0 1; do
rm -rf $R/gpurun_out/sp$v
UNINA_STEM_V1=$v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sp$v -o t --output-format csv -- python3 $R/tools/profile_ops.py > /dev/null 2>&1
grep -i "stem" $R/gpurun_out/sp$v/*/t_kernel_stats.csv $R/gpurun_out/sp$v/t_kernel_stats.csv 2>/dev/null | cut -c1-220
done
