mkdir -p gpurun_out/minch
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wgrad or batched_parameter_work or backward_against_autograd_oracle or reduce_the_loss or data_parallel or captured" > gpurun_out/minch/t.log 2>&1
tail -3 gpurun_out/minch/t.log
