cd /root/repo
rm -rf /tmp/prof_r05
timeout 2400 bash tools/prof_all.sh /tmp/prof_r05 > /tmp/prof_r05.log 2>&1
tail -2 /tmp/prof_r05.log
python tools/prof_report.py /tmp/prof_r05 gpurun_out/r05a 2>&1 | tail -3
cp /tmp/prof_r05/*.log gpurun_out/ 2>/dev/null; rm -f gpurun_out/*_sq*.log gpurun_out/*_fetch.log gpurun_out/*_write.log
ls -la gpurun_out | grep r05a | wc -l
