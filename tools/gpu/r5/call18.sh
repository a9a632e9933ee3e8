#!/bin/bash
# round 5, call 18: the S5 sweep loop (100 launches per frame, no host decision inside) as ONE hipGraph per frame: parity, alone, in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(TMC2_REFINE_GRAPH=1 timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py -m gpu -x -q -k "refine or segmenter_compute" 2>&1 | tail -4) > $O/r05c18_graph_tests.log 2>&1; tail -1 $O/r05c18_graph_tests.log
for g in 0 1; do
TMC2_REFINE_GRAPH=$g timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c18_graph$g.json 2> $O/r05c18_graph$g.err
python -c "
import json; d=json.loads(open('$O/r05c18_graph$g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('graph $g', d['value'], d['verified'], 'proxy', d.get('per_rank_proxy',{}).get('ms'), 'sweeps in flight', d['stage_ms_per_frame']['refine_sweeps'], 'alone', r['stages']['refine_sweep']['alone_ms'])"
done
