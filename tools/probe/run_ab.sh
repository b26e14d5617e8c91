mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests -m gpu -x -q -k "pending_releases or terminal_observation" > gpurun_out/r2m/pytest.txt 2>&1; tail -c 2500 gpurun_out/r2m/pytest.txt
