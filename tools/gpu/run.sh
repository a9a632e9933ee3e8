cd /root/repo
export TMPDIR=/tmp
bash tools/gpu/smi_sample.sh gpurun_out/r02_smi_default.jsonl timeout 300 python bench.py --steps 24 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_smi_bench.json 2> gpurun_out/r02_smi_bench.err
bash tools/gpu/smi_sample.sh gpurun_out/r02_smi_solo.jsonl timeout 300 python bench.py --steps 6 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_smi_solo.json 2> gpurun_out/r02_smi_solo.err
