cd /root/repo
for c in "" text elf json image; do
  echo "== corpus:$c"
  NXZ_ENGINE_LIB=libnxz_engine_prof.so FC=dhtgen timeout 300 python tools/phase_profile.py 4096 corpus:$c 2>&1 | tail -19
done > gpurun_out/r05_phase_by_class_after.txt 2>&1
