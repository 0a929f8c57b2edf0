# the round's measurements in one call on the GPU box: the default bench line, the profiler passes of the legs whose kernels changed
# (summaries only come back: the raw rocprofv3 directories are deleted), the API sweep, the long-stream table
mkdir -p gpurun_out/r06d
python bench.py > gpurun_out/r06d/bench_default.json 2> gpurun_out/r06d/bench_default.err
bash tools/prof_all.sh gpurun_out/r06d/prof dhtgen inflate_zlib6 inflate_own > gpurun_out/r06d/prof_all.log 2>&1
python tools/prof_report.py gpurun_out/r06d/prof gpurun_out/r06d/r06 >> gpurun_out/r06d/prof_all.log 2>&1
find gpurun_out/r06d/prof -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
SWEEP_KIB=4,16,64,128,256,512,1024,4096,16384 python tools/api_sweep.py > gpurun_out/r06d/api_sweep.txt 2>&1
python tools/lab/bench_wg_long.py > gpurun_out/r06d/long_streams.txt 2>&1
du -sh gpurun_out; tail -c 600 gpurun_out/r06d/bench_default.json; tail -3 gpurun_out/r06d/prof_all.log
