mkdir -p gpurun_out/soak
timeout -k 10 900 python tests/dev_soak.py > gpurun_out/soak/soak.log 2>&1; tail -15 gpurun_out/soak/soak.log
