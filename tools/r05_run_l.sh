#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
python3 tools/exp_two_streams.py > $O/two.json 2> $O/two.err
SSW_SEN_GROUPS=512 python3 tools/exp_two_streams.py > $O/two_g512.json 2>> $O/two.err
SSW_SEN_GROUPS=768 python3 tools/exp_two_streams.py > $O/two_g768.json 2>> $O/two.err
python3 tools/exp_two_streams.py --frames 16384 --steps 200 > $O/two_16k.json 2>> $O/two.err
python3 tools/exp_two_streams.py --models 3 > $O/three.json 2>> $O/two.err
python3 tools/bench_host_boundary.py > $O/host_boundary.json 2>> $O/two.err
bash tools/bench_sen_shapes.sh r05l > $O/shapes.log 2>&1
tail -3 $O/two.err
for f in two two_g512 two_g768 two_16k three; do echo $f; cat $O/$f.json; done
python3 -c "
import json; j=json.load(open('$O/host_boundary.json')); print({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if 'frames_per_s' in a or 'ragged' in a}) for k,v in j.items() if k!='note'})"
cat $O/sen_shapes.json
