#!/bin/bash
# the three soaks the closing call's 60-minute limit cut off (same sources)
tag=r05; secs=${1:-300}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python tests/soak_parity.py --mode first_pass --seconds $secs > $O/${tag}_parity_soak_first_pass.json 2> $O/${tag}_soak_first_pass.err
python tests/soak_parity.py --mode first_pass --model fr-fr --seconds $secs > $O/${tag}_parity_soak_first_pass_frfr.json 2> $O/${tag}_soak_first_pass_frfr.err
python tests/soak_parity.py --mode text --seconds $secs > $O/${tag}_parity_soak_text.json 2> $O/${tag}_soak_text.err
tail -n 2 $O/${tag}_parity_soak_first_pass*.json $O/${tag}_parity_soak_text.json
