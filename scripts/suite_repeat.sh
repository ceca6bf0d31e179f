mkdir -p gpurun_out/suite3
for i in 1 2 3 4; do
  (time timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -8) > gpurun_out/suite3/run$i.log 2>&1
done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/suite3/smoke.log 2>&1
python bench.py > gpurun_out/suite3/bench.json 2> gpurun_out/suite3/bench.err
for f in gpurun_out/suite3/run*.log; do tail -3 "$f"; done
