#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
R=/root/repo
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
  echo -n "$lib: "; VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python scripts/tiled_pass.py 4 2>&1 | grep -v amdgpu.ids
done
done | tee $O/r06_tiled_ab.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/r06_prof_t -o tiled -- python $R/scripts/tiled_pass.py 3 > $R/$O/r06_prof_t.log 2>&1); echo "rocprof tiled rc=$?"
DB=$(find $O/r06_prof_t -name "*.db" | head -1); rm -f $O/r06_tiled_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/r06_tiled_kernel_stats.md; grep -i "lerp\|gather\|layernorm" $O/r06_tiled_kernel_stats.md | cut -c1-170
rm -rf $O/r06_prof_t
