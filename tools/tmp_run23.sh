cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpus.py tests/test_gpu_dhtgen.py tests/test_gpu_deflate_batch.py tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -4
for f in 1 0; do
NXZ_FUSED_GEN=$f timeout 900 python bench.py --no-inflate --no-api --no-c2 --no-c5 --no-cpu-baseline > gpurun_out/r05_bench_fused$f.json 2> gpurun_out/r05_bench_fused$f.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05_bench_fused$f.json'))
c=d['config']
print("NXZ_FUSED_GEN=$f", d['value'], d['ms_per_step'], c['ratio_vs_zlib1'], c.get('min_class_GiB_s'), c.get('silesia_weighted_GiB_s',{}).get('value'), {k:v.get('GiB_s') for k,v in c['classes'].items()}, d['roofline']['lz77_ms'], d['roofline']['dhtgen_ms'], d['roofline']['entropy_ms'], d['roofline']['launches_per_step'])
PY
done
