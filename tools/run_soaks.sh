#!/bin/bash
# Randomised GPU-vs-oracle soaks on the current build (ON the GPU box):
#   gpurun --timeout 3000 -- 'bash tools/run_soaks.sh r04 240'
# Writes gpurun_out/<tag>_parity_soak_*.json (copy into profiles/ to keep them).
tag=${1:-r06}
secs=${2:-240}
seed=${3:-0}   # other inputs than the default run's: pass 1, 2, ...
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python tests/soak_parity.py --seed $seed --mode ptm --seconds $secs > $O/${tag}_parity_soak_ptm.json 2> $O/${tag}_soak_ptm.err
# the same with every third wave of the matrix-core scan audited in-kernel (and every wave)
SSW_SCAN_AUDIT=3 python tests/soak_parity.py --seed $seed --mode ptm --seconds $((secs / 2)) > $O/${tag}_parity_soak_ptm_audit3.json 2> $O/${tag}_soak_ptm_audit3.err
SSW_SCAN_AUDIT=1 python tests/soak_parity.py --seed $seed --mode ptm --max-len 3000 --seconds $((secs / 2)) > $O/${tag}_parity_soak_ptm_audit1_long.json 2> $O/${tag}_soak_ptm_audit1_long.err
SSW_SCAN_AUDIT=2 python tests/soak_parity.py --seed $seed --mode ms --model fr-fr --seconds $((secs / 2)) > $O/${tag}_parity_soak_ms_audit2.json 2> $O/${tag}_soak_ms_audit2.err
python tests/soak_parity.py --seed $seed --mode ptm --model fr-fr --seconds $((secs / 2)) > $O/${tag}_parity_soak_ptm_frfr.json 2> $O/${tag}_soak_ptm_frfr.err
python tests/soak_parity.py --seed $seed --mode ms --model fr-fr --seconds $secs > $O/${tag}_parity_soak_ms.json 2> $O/${tag}_soak_ms.err
python tests/soak_parity.py --seed $seed --mode align --seconds $secs > $O/${tag}_parity_soak_align.json 2> $O/${tag}_soak_align.err
python tests/soak_parity.py --seed $seed --mode topo --seconds $((secs / 2)) > $O/${tag}_parity_soak_topo.json 2> $O/${tag}_soak_topo.err
python tests/soak_parity.py --seed $seed --mode first_pass --seconds $secs > $O/${tag}_parity_soak_first_pass.json 2> $O/${tag}_soak_first_pass.err
python tests/soak_parity.py --seed $seed --mode first_pass --model fr-fr --seconds $secs > $O/${tag}_parity_soak_first_pass_frfr.json 2> $O/${tag}_soak_first_pass_frfr.err
python tests/soak_parity.py --seed $seed --mode text --seconds $secs > $O/${tag}_parity_soak_text.json 2> $O/${tag}_soak_text.err
# round 6: the default configuration's first pass as a batch against the frame-synchronous oracle
python tests/soak_parity.py --seed $seed --mode fp_active --seconds $secs > $O/${tag}_parity_soak_fp_active.json 2> $O/${tag}_soak_fp_active.err
python tests/soak_parity.py --seed $seed --mode text_active --seconds $((secs / 2)) > $O/${tag}_parity_soak_text_active.json 2> $O/${tag}_soak_text_active.err
SSW_FPA_SUB=1 python tests/soak_parity.py --seed $((seed + 100)) --mode fp_active --seconds $((secs / 2)) > $O/${tag}_parity_soak_fp_active_one_by_one.json 2> $O/${tag}_soak_fp_active_one_by_one.err
tail -n 2 $O/${tag}_parity_soak_*.json
tail -n 3 $O/${tag}_soak_*.err
