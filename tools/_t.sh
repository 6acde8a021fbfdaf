mkdir -p gpurun_out/c4
for f in test_slabs test_slabs_multiprocess test_particle_parity test_fv_known_answers_gpu test_fibre_coupling test_locate_paths; do
  timeout 900 python -m pytest tests/$f.py -x -q -m gpu --durations=6 > gpurun_out/c4/$f.log 2>&1; echo "$f rc=$?" >> gpurun_out/c4/rc.log
done
cat gpurun_out/c4/rc.log
for f in gpurun_out/c4/test_*.log; do echo "== $f"; tail -12 $f; done
