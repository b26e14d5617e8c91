mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2m/pytest.txt 2>&1; tail -c 600 gpurun_out/r2m/pytest.txt
