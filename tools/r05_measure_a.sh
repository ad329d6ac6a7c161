#!/bin/bash
# Round 5, first measurement call (ON the GPU box): the default-configuration batch path timed
# and profiled, and PMC passes over the alignment kernel at 256 and 2048 utterances.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python3 tools/bench_align_active.py > $O/align_active.json 2> $O/align_active.err
timeout 600 python3 tools/bench_align_active.py --ms > $O/align_active_ms.json 2> $O/align_active_ms.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_active -o active -- python3 $R/tools/bench_align_active.py --reps 2 > $O/active_under_rocprof.json 2> $O/active_rocprof.err
find $O/prof_active -name "*kernel_stats.csv" -exec cp {} $O/align_active_kernel_stats.csv \;
find $O/prof_active -name "*kernel_trace.csv" -delete
cd $R
timeout 900 python3 tools/pmc_cmd.py gpurun_out/r05a/align_pmc_256.json 'viterbi_align\w+|ptm_senone_kernel|ptm_topn_mfma_kernel' -- python3 tools/bench_align.py --utts 256 --reps 1 > $O/pmc256.log 2>&1
timeout 1500 python3 tools/pmc_cmd.py gpurun_out/r05a/align_pmc_2048.json 'viterbi_align\w+|ptm_senone_kernel|ptm_topn_mfma_kernel' -- python3 tools/bench_align.py --utts 2048 --reps 1 > $O/pmc2048.log 2>&1
rm -rf $R/gpurun_out/pmc_align_pmc_256 $R/gpurun_out/pmc_align_pmc_2048
ls -la $O
tail -3 $O/pmc256.log $O/pmc2048.log
cat $O/align_active.json $O/align_active_ms.json
