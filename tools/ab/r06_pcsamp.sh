# r06: instruction-level evidence for the sweep kernels (gpurun -- 'bash tools/ab/r06_pcsamp.sh [C3] [interval]').
#  1. rocprofv3 --att (thread trace): this image ships no trace decoder (no librocprof-trace-decoder.so under /opt/rocm) -- the attempt and
#     its message are recorded so that the absence is a measured fact, not an assumption.
#  2. stochastic PC sampling (hardware sampler of gfx950: per sample the wave's PC, whether it issued, the reason it did not, and the
#     arbiter's per-pipe issue / stall state) of ONE sweep (tools/pmc_sweep.py), kernel-trace beside it for the dispatch -> kernel map.
# Never combined with --pmc or the hip/hsa/sys traces.  Every call under its own timeout.
W=${1:-C3}
IV=${2:-1048576}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_pcsamp_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${SKIP_ATT:-0}" != "1" ]; then
  timeout 180 rocprofv3 --att --kernel-trace --kernel-include-regex 'k_contract16' --output-format csv -d $OUT/att -o att -- python $ROOT/tools/pmc_sweep.py $W > $OUT/att.log 2>&1
  echo "att rc=$?" >> $OUT/att.log
  find $OUT/att -type f | head -20 >> $OUT/att.log
  rm -rf $OUT/att
fi
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval $IV \
  --kernel-trace --output-format csv json -d $OUT/pcs -o pcs -- python $ROOT/tools/pmc_sweep.py $W > $OUT/pcs.log 2>&1
echo "pcs rc=$?" >> $OUT/pcs.log
find $OUT/pcs -type f -exec ls -la {} \; >> $OUT/pcs.log
cd $ROOT
python tools/pcsamp_summary.py $OUT/pcs > $OUT/summary.txt 2> $OUT/summary.err
for f in $(find $OUT/pcs -type f -name "*.csv" -o -type f -name "*.json"); do
  head -c 6000 $f > $OUT/head_$(basename $f).txt
  sz=$(stat -c %s $f)
  if [ $sz -lt 400000000 ]; then gzip -1 -c $f > $OUT/$(basename $f).gz; fi
done
rm -rf $OUT/pcs
# drop anything that would overflow the 64-MiB merge
find $OUT -name "*.gz" -size +28M -delete
du -sh $OUT; ls -la $OUT
