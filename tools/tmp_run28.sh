cd /root/repo
timeout 1800 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
timeout 900 python bench.py --config c2 > gpurun_out/r05_bench_c2.json 2>> gpurun_out/r05_bench_default.err
timeout 900 python bench.py --config c5 > gpurun_out/r05_bench_c5.json 2>> gpurun_out/r05_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_default.json'))
c=d['config']
print("headline", d['value'], d['ms_per_step'], "ratio", c['ratio'], c['ratio_vs_zlib1'], c['min_class_vs_zlib1'], "minclass", c['min_class_GiB_s'], "silesia", c['silesia_weighted_GiB_s']['value'])
print({k:(v.get('GiB_s'), v['vs_zlib1']) for k,v in c['classes'].items()})
print("roofline", {k:d['roofline'][k] for k in ('achieved','frac','traffic','lz77_ms','dhtgen_ms','entropy_ms','launches_per_step','frac_of_measured')})
print("cpu", d['cpu_baseline']['value'], d['cpu_baseline']['one_thread_GiB_s'])
for k in ('inflate_zlib6','inflate_stream'):
    print(k, d[k]['value'], d[k].get('ms_per_pass', d[k].get('ms')), d[k].get('without_image_and_packed_classes'))
print("api", {k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in v if kk in ('value','first_calls_ms','us_per_call')}) for k,v in d['api'].items() if k!='note'})
print("c2", d['c2']['value'], d['c2']['ms_per_step'], {k:d['c2'][k]['value'] for k in d['c2'] if isinstance(d['c2'][k],dict) and 'value' in d['c2'][k]})
print("c5", d['c5']['value'], d['c5']['ms_per_step'])
for f in ('c2','c5'):
    x=json.load(open('gpurun_out/r05_bench_%s.json'%f)); print(f, "own line:", x['value'], x['ms_per_step'])
PY
