cd /root/repo
timeout 900 python -m pytest tests/test_gpu_dhtgen.py tests/test_gpu_corpus.py -x -q -m gpu 2>&1 | tail -3
NXZ_FUSED_GEN=1 timeout 900 python -m pytest tests/test_gpu_dhtgen.py tests/test_gpu_corpus.py tests/test_gpu_parity.py -x -q -m gpu -k "not inflate" 2>&1 | tail -3
for f in 1 0; do
NXZ_FUSED_GEN=$f timeout 900 python bench.py --no-inflate --no-api --no-c2 --no-c5 --no-cpu-baseline --class-jobs 32768 > gpurun_out/r05_bench_fused$f.json 2> gpurun_out/r05_bench_fused$f.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05_bench_fused$f.json'))
c=d['config']
print("NXZ_FUSED_GEN=$f", d['value'], d['ms_per_step'], d['roofline']['lz77_ms'], d['roofline']['dhtgen_ms'], d['roofline']['entropy_ms'], {k:v.get('GiB_s') for k,v in c['classes'].items()})
PY
done
