# r06: which PC-sampling configurations does this box offer?  (gpurun -- 'bash tools/ab/r06_pcsamp_probe.sh')
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_pcsamp_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 60 rocprofv3-avail list --pc-sampling > $OUT/avail_list.txt 2>&1
timeout 60 rocprofv3-avail info --pc-sampling > $OUT/avail_info.txt 2>&1
cat /sys/module/amdgpu/version > $OUT/driver.txt 2>&1; uname -r >> $OUT/driver.txt
# host-trap, time unit (us)
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 \
  --kernel-trace --output-format csv -d $OUT/ht -o ht -- python $ROOT/tools/pmc_sweep.py C3 > $OUT/ht.log 2>&1
echo "ht rc=$?" >> $OUT/ht.log
find $OUT/ht -type f -exec ls -la {} \; >> $OUT/ht.log
cd $ROOT
python tools/pcsamp_summary.py $OUT/ht > $OUT/ht_summary.txt 2> $OUT/ht_summary.err
for f in $(find $OUT/ht -type f -name "*.csv"); do head -c 4000 $f > $OUT/head_$(basename $f).txt; gzip -1 -c $f > $OUT/$(basename $f).gz; done
rm -rf $OUT/ht
find $OUT -name "*.gz" -size +28M -delete
ls -la $OUT
