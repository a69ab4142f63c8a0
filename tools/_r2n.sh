mkdir -p gpurun_out/r2n
python -m pytest tests -m gpu -x -q > gpurun_out/r2n/tests.log 2>&1; tail -3 gpurun_out/r2n/tests.log
( python tools/sweep.py showcase 4 "" merged=0 steal=0
  python tools/sweep.py cornell 4 ""
  python tools/sweep.py fluid 2 "" merged=0
  python tools/sweep.py many 4 "" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n/out.txt
