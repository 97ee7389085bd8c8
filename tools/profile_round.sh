#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command + PMC passes over the kernel table.
# (counters in their own runs with --kernel-trace only; the program itself after `--`)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r03}
export TDVC_PROFILE_TAG=$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
set -x
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table > $OUT/prof_$TAG.log 2>&1 || exit 1
timeout -k 10 300 python3 $R/tools/microbench_kernels.py --top 40 --ops-file $OUT/pmc_ops.json > $OUT/pmc_select_$TAG.log 2>&1 || exit 1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$name -- python3 $R/tools/microbench_kernels.py --iters 6 --ops-file $OUT/pmc_ops.json --manifest $OUT/pmc_manifest_$name.json > $OUT/pmc_${TAG}_$name.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_traffic.py $OUT/pmc_manifest_fetch.json $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_sq > $OUT/pmc_$TAG.txt 2>&1
tail -5 $OUT/pmc_$TAG.txt
# gpurun copies at most 64 MiB back: keep the summaries, compress the step trace, drop the raw counter dumps
gzip -f $OUT/prof_$TAG/*/*_kernel_trace.csv 2>/dev/null
rm -rf $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_sq
