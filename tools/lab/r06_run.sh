# the round's measurements in one call on the GPU box: the GPU suite, the default bench line, the profiler passes of the legs whose
# kernels or routes changed (summaries only come back: the raw rocprofv3 directories are deleted), the sweeps
mkdir -p gpurun_out/r06k
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r06k/pytest_gpu.txt
python bench.py > gpurun_out/r06k/bench_default.json 2> gpurun_out/r06k/bench_default.err
bash tools/prof_all.sh gpurun_out/r06k/prof inflate_wg > gpurun_out/r06k/prof_all.log 2>&1
python tools/prof_report.py gpurun_out/r06k/prof gpurun_out/r06k/r06 >> gpurun_out/r06k/prof_all.log 2>&1
find gpurun_out/r06k/prof -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
KERNELS=wg,old SIZES=64,256,1024,4096,16384,65536,131072,262144 python tools/bench_inflate_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06k/inflate_by_batch_size.txt
SWEEP_KIB=4,16,64,128,256,512,1024,4096,16384 python tools/api_sweep.py > gpurun_out/r06k/api_sweep.txt 2>&1
python tools/lab/bench_wg_long.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06k/long_streams.txt
cat gpurun_out/r06k/pytest_gpu.txt; du -sh gpurun_out; tail -3 gpurun_out/r06k/prof_all.log
