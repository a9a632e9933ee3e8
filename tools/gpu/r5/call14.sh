#!/bin/bash
# round 5, call 14: the vox11 configuration (encoder side + decoder-side leg) with the new tree build and with round 4's tiers
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for form in pieces tiers; do
TMC2_KD_FORM=$form timeout -k 10 900 python bench.py --config basketball --steps 4 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > $O/r05c14_basketball_$form.json 2> $O/r05c14_basketball_$form.err; echo "rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c14_basketball_$form.json').read().strip().splitlines()[-1]); dec=d.get('decoder',{})
print('$form', d['value'], d['verified'], 'proxy', d.get('per_rank_proxy',{}).get('ms'), 'decoder', dec.get('frames_per_s'), dec.get('verified'))
print('  enc stages', {k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k}, 'metric alone', d.get('metric_ms_per_frame'), d.get('metric_stage_ms'))
print('  dec stages', sorted(dec.get('stage_ms_per_frame',{}).items(), key=lambda kv:-kv[1])[:8])"
done
