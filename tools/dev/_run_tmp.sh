mkdir -p gpurun_out/final2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final2/gpu_tests.log 2>&1; tail -3 gpurun_out/final2/gpu_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
