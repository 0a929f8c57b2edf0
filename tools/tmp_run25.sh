cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05_t25.txt
cat gpurun_out/r05_t25.txt
rm -rf gpurun_out/prof_r05
timeout 2400 bash tools/prof_all.sh gpurun_out/prof_r05 > gpurun_out/prof_r05.log 2>&1
tail -3 gpurun_out/prof_r05.log
python tools/prof_report.py gpurun_out/prof_r05 gpurun_out/r05a 2>&1 | tail -5
ls gpurun_out/ | grep r05a | head -20
# keep what is small enough to travel
rm -rf gpurun_out/prof_r05/*/*/*.db gpurun_out/prof_r05/*/*.db 2>/dev/null
du -sh gpurun_out/prof_r05 | tail -1
